"""-m gpu: the opt-in consistent tangent (FEMCY_OPT_TANGENT = 1).  There is no reference oracle for
it (the reference only has the commented-out hooks, neo_hookean.py:62-64, 79-81): it is checked against what it
must be, the derivative of the internal force -- K v = d/dh f_int(u + h v) by central differences -- and by what it
buys, Newton converging on the same equilibrium in far fewer linear solves."""
import numpy as np
import pytest

from helpers import deck

pytestmark = pytest.mark.gpu


def load(name):
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    et = list(inp.eSets)[0]
    return inp, inp.eSets[et], list(inp.materials.values())[0]


def smooth_disp(nodes, scale):
    L = np.ptp(nodes, axis=0).max()
    x = nodes / L
    u = np.stack([np.sin(1.3 * x[:, 0] + 0.4) * np.cos(0.7 * x[:, -1]), 0.5 * np.cos(2.1 * x[:, 1] - 0.2) * x[:, 0],
                  0.3 * np.sin(x.sum(axis=1))][:nodes.shape[1]], axis=1)
    return (scale * L * u).ravel()


@pytest.mark.parametrize("name", ["twist_plate_C3D4.inp",                  # StVK 3-D
                                  "cook_3d_linearEl_largeDef.inp",         # neo-Hookean
                                  "twist_C3D10_coarse.inp",                # 4 Gauss points
                                  "cookMembrane_2d_linearEl_largeDef.inp"])   # plane strain
def test_tangent_is_the_derivative_of_the_internal_force(gpu_ctx_factory, name):
    from femcy_amd import backend as be
    inp, el, mat = load(name)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(inp.nodes, el)
    ctx.set_element(inp.ELE)
    ctx.set_material(mat)
    ctx.build_pattern()
    u = smooth_disp(inp.nodes, 0.05)                       # ~5 % strains: far from the linear regime
    rng = np.random.default_rng(3)
    L = np.ptp(inp.nodes, axis=0).max()

    def f_int(w):
        ctx.upload(be.VEC_DOF, w)
        ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
        return ctx.download(be.VEC_FORCE)

    def Kv(v):
        ctx.upload(be.VEC_TMP0, v)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        return ctx.download(be.VEC_TMP1)

    errs = {}
    for tangent in (1, 0):
        ctx.set_option(be.OPT_TANGENT, tangent)
        ctx.upload(be.VEC_DOF, u)
        ctx.assemble_K(be.VEC_DOF)
        worst = 0.0
        for _ in range(3):
            v = rng.standard_normal(u.size)
            h = 1e-6 * L
            fd = (f_int(u + h * v) - f_int(u - h * v)) / (2 * h)
            worst = max(worst, np.linalg.norm(Kv(v) - fd) / np.linalg.norm(fd))
        errs[tangent] = worst
        if tangent == 1:                                       # symmetric, like every hyperelastic tangent
            x, y = rng.standard_normal(u.size), rng.standard_normal(u.size)
            assert abs(y @ Kv(x) - x @ Kv(y)) <= 1e-10 * abs(y @ Kv(x))
    assert errs[1] < 2e-7, errs                # central differences: O(h^2) + rounding/h
    assert errs[0] > 1e-3, errs                # the reference's matrix is NOT the derivative (modified Newton)
    # small-strain limit: both coincide with B^T C B at u = 0 (neo-Hookean: with its own linearisation)
    if mat.kind != 3:
        ctx.set_option(be.OPT_TANGENT, 1)
        ctx.assemble_K(-1)
        y1 = Kv(u)
        ctx.set_option(be.OPT_TANGENT, 0)
        ctx.assemble_K(-1)
        assert np.linalg.norm(y1 - Kv(u)) <= 1e-12 * np.linalg.norm(y1)


def test_plane_stress_is_rejected(gpu_ctx_factory):
    from femcy_amd import backend as be
    inp, el, mat = load("beam_CPS3_disp_meshSize5.inp")
    ctx = gpu_ctx_factory()
    ctx.set_mesh(inp.nodes, el)
    ctx.set_element(inp.ELE)
    ctx.set_material(mat)
    ctx.build_pattern()
    ctx.set_option(be.OPT_TANGENT, 1)
    with pytest.raises(be.FemcyError, match="plane stress"):
        ctx.assemble_K(-1)
    with pytest.raises(be.FemcyError):
        ctx.set_option(be.OPT_TANGENT, 2)


@pytest.mark.parametrize("name,solve_gain,eval_gain", [("twist_plate_C3D4.inp", 3.0, 2.5),
                                                        ("cook_3d_linearEl_largeDef.inp", 1.0, 1.6)])
def test_newton_with_the_consistent_tangent(name, solve_gain, eval_gain):
    """same deck, same increments logic: the consistent tangent reaches the same equilibrium (the reference's
    convergence test is relative residual < 1 %, so the two end states agree to about that) with a fraction of
    the linear solves."""
    from femcy_amd.body import Body
    from femcy_amd.stiffnessMtrx import System_of_equations
    inp, el, mat = load(name)
    out = {}
    for tangent in ("reference", "consistent"):
        system = System_of_equations(Body(inp.nodes, el, inp.ELE), mat, inp.geometric_nonlinear, verbose=False,
                                     tangent=tangent)
        system.solve(inp)
        out[tangent] = (system.dof.to_numpy(), dict(system.stats), system.increments)
        assert system.time0 == inp.time_incs["max_time"]
        system.ctx.close()
    (u0, s0, _), (u1, s1, _) = out["reference"], out["consistent"]
    assert np.linalg.norm(u1 - u0) <= 2e-2 * np.linalg.norm(u0)
    print(name, "reference", s0, "consistent", s1)
    assert s1["linear_solves"] * solve_gain <= s0["linear_solves"], (s0, s1)
    assert s1["force_evals"] * eval_gain <= s0["force_evals"], (s0, s1)     # line-search evaluations mostly vanish
