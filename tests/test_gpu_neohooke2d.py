"""-m gpu: plane-strain neo-Hookean on 2-D elements (BASELINE configs[1] names "CPE8 large-def Neo-Hookean"; the
reference has neo-Hookean for 3-D elements only and rejects it on CPS/CPE decks, so this is an EXTENSION with no
reference oracle).  It is verified against the 3-D class, which IS checked against the oracle: a one-layer extrusion
of the 2-D mesh into tetrahedra with u_z = 0 is the same plane-strain problem."""
import numpy as np
import pytest

from helpers import deck

pytestmark = pytest.mark.gpu

C1, D1 = 3.85e4, 8.3e4


def extrude(nodes2, tris):
    """unit-thickness slab of the triangle mesh: each triangle -> prism -> 3 tetrahedra, oriented for the reference's
    C3D4 shape functions N = [zeta, xi, 1 - xi - eta - zeta, eta] (det J > 0)."""
    nn = nodes2.shape[0]
    nodes3 = np.concatenate([np.c_[nodes2, np.zeros(nn)], np.c_[nodes2, np.ones(nn)]])
    dN = np.array([[0., 0., 1.], [1., 0., 0.], [-1., -1., -1.], [0., 1., 0.]])
    tets, parent = [], []
    for e, (a, b, c) in enumerate(tris):
        A, B, C = a + nn, b + nn, c + nn
        for t in ((a, b, c, A), (b, c, A, B), (c, A, B, C)):
            t = list(t)
            if np.linalg.det(nodes3[t].T @ dN) < 0:
                t[0], t[1] = t[1], t[0]
            assert np.linalg.det(nodes3[t].T @ dN) > 0
            tets.append(t)
            parent.append(e)
    return nodes3, np.array(tets, dtype=np.int32), np.array(parent)


def test_plane_strain_equals_constrained_slab(gpu_ctx_factory):
    from femcy_amd import backend as be
    from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_linear_triangular
    from femcy_amd.material_zoo import NeoHookean, NeoHookeanPlaneStrain
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck("cookMembrane_2d_linearEl_largeDef.inp"))          # CPE3 mesh of Cook's membrane
    nodes2, tris = inp.nodes, inp.eSets["CPE3"]
    nn = nodes2.shape[0]
    L = np.ptp(nodes2, axis=0).max()
    x = nodes2 / L
    u2 = 0.08 * L * np.stack([np.sin(1.3 * x[:, 0] + 0.4) * np.cos(0.7 * x[:, 1]), 0.5 * np.cos(2.1 * x[:, 1]) * x[:, 0]], 1)

    c2 = gpu_ctx_factory()
    c2.set_mesh(nodes2, tris)
    c2.set_element(Element_linear_triangular())
    c2.set_material(NeoHookeanPlaneStrain(C1, D1))
    c2.build_pattern()
    c2.upload(be.VEC_DOF, u2.ravel())
    c2.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f2 = c2.download(be.VEC_FORCE).reshape(-1, 2)
    F2, S2 = c2.gauss_field(be.GP_F).to_numpy()[:, 0], c2.gauss_field(be.GP_SIGMA).to_numpy()[:, 0]
    e2 = c2.elastic_energy(be.VEC_DOF)
    c2.compute_strain_stress(be.VEC_DOF, large=True)
    m2 = c2.gauss_field(be.GP_MISES).to_numpy()[:, 0]

    nodes3, tets, parent = extrude(nodes2, tris)
    u3 = np.zeros((2 * nn, 3))
    u3[:nn, :2] = u3[nn:, :2] = u2
    c3 = gpu_ctx_factory()
    c3.set_mesh(nodes3, tets)
    c3.set_element(Element_linear_tetrahedral())
    c3.set_material(NeoHookean(C1, D1))
    c3.build_pattern()
    c3.upload(be.VEC_DOF, u3.ravel())
    c3.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f3 = c3.download(be.VEC_FORCE).reshape(-1, 3)
    F3, S3 = c3.gauss_field(be.GP_F).to_numpy()[:, 0], c3.gauss_field(be.GP_SIGMA).to_numpy()[:, 0]
    e3 = c3.elastic_energy(be.VEC_DOF)
    c3.compute_strain_stress(be.VEC_DOF, large=True)
    m3 = c3.gauss_field(be.GP_MISES).to_numpy()[:, 0]

    assert np.abs(F3[:, :2, :2] - F2[parent]).max() < 1e-13 and np.abs(F3[:, 2, 2] - 1.0).max() < 1e-13
    scale = np.abs(S2).max()
    assert np.abs(S3[:, :2, :2] - S2[parent]).max() < 1e-12 * scale            # in-plane Cauchy stress
    assert np.abs(f3[:nn, :2] + f3[nn:, :2] - f2).max() < 1e-11 * np.abs(f2).max()      # both layers carry one plate
    assert abs(e3 - e2) < 1e-11 * abs(e2)
    assert np.abs(m3 - m2[parent]).max() < 1e-11 * m2.max()                    # von Mises incl. the out-of-plane stress


def test_tangent_is_the_derivative_2d(gpu_ctx_factory):
    from femcy_amd import backend as be
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    from femcy_amd.material_zoo import NeoHookeanPlaneStrain
    from femcy_amd import meshgen
    m = meshgen.beam_quad8(10, 2, plane="CPE8", tip_disp=4.0)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_quadrilateral())
    ctx.set_material(NeoHookeanPlaneStrain(C1, D1))
    ctx.build_pattern()
    ctx.set_option(be.OPT_TANGENT, 1)
    L = 40.0
    x = m["nodes"] / L
    u = (0.05 * L * np.stack([np.sin(1.3 * x[:, 0] + 0.4), 0.5 * np.cos(2.1 * x[:, 1]) * x[:, 0]], 1)).ravel()
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)
    rng = np.random.default_rng(1)

    def f_int(w):
        ctx.upload(be.VEC_DOF, w)
        ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
        return ctx.download(be.VEC_FORCE)

    for _ in range(3):
        v = rng.standard_normal(u.size)
        h = 1e-6 * L
        fd = (f_int(u + h * v) - f_int(u - h * v)) / (2 * h)
        ctx.upload(be.VEC_TMP0, v)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        assert np.linalg.norm(ctx.download(be.VEC_TMP1) - fd) < 2e-7 * np.linalg.norm(fd)


def test_cpe8_beam_neo_hookean_end_to_end():
    """BASELINE configs[1] as written: CPE8, large deformation, neo-Hookean, through the whole driver; both tangents
    arrive at the same equilibrium (the reference's criterion is a 1 % residual)."""
    from types import SimpleNamespace
    from femcy_amd import meshgen
    from femcy_amd.body import Body
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    from femcy_amd.material_zoo import NeoHookeanPlaneStrain
    from femcy_amd.stiffnessMtrx import System_of_equations
    m = meshgen.beam_quad8(20, 2, plane="CPE8", tip_disp=4.0)
    ELE = Element_quadratic_quadrilateral()
    inp = SimpleNamespace(dirichlet_bc_info=m["dirichlet_bc_info"], neumann_bc_info=[], time_incs=m["time_incs"],
                          geometric_nonlinear=True)
    out = {}
    for tangent in ("reference", "consistent"):
        system = System_of_equations(Body(m["nodes"], m["elements"], ELE), NeoHookeanPlaneStrain(C1, D1), True,
                                     verbose=False, tangent=tangent)
        system.solve(inp)
        assert system.time0 == 1.0 and all(i["converged"] for i in system.increments)
        u = system.dof.to_numpy().reshape(-1, 2)
        right = m["node_sets"]["right_side"]
        assert np.allclose(u[right, 1], 4.0) and np.allclose(u[right, 0], 0.0)       # prescribed tip displacement
        out[tangent] = (u, system.get_elasEng(), dict(system.stats))
        system.ctx.close()
    (u0, e0, s0), (u1, e1, s1) = out["reference"], out["consistent"]
    assert np.linalg.norm(u1 - u0) <= 2e-2 * np.linalg.norm(u0) and abs(e1 - e0) <= 3e-2 * e0 and e0 > 0
    assert s1["linear_solves"] <= s0["linear_solves"]
    print("CPE8 neo-Hookean beam:", s0, s1)
