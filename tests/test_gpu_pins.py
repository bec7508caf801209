"""-m gpu: the HIP path against pins that come from neither the oracle's nor the product's tables.

  * one-element meshes against the exactly integrated K^e and the closed-form homogeneous-deformation forces of
    tests/sympy_pins.py (derived symbolically from the node ordering alone);
  * affine patch tests on multi-element meshes: under a homogeneous deformation gradient every Gauss point must carry
    the closed-form Cauchy stress (written out here, in the test) and every interior node must be in equilibrium;
  * the opt-in consistent tangent against the oracle's complex-step derivative of the oracle's internal force
    (oracle.consistent_tangent) -- before round 2 that kernel was only ever compared with finite differences of the
    HIP force itself.
"""
import numpy as np
import pytest

from helpers import deck, oracle_material
from oracle import femcy_oracle as orc
from oracle.elements import elem_def

pytestmark = pytest.mark.gpu


def _ctx(factory, nodes, el, ELE, mat):
    ctx = factory()
    ctx.set_mesh(nodes, el)
    ctx.set_element(ELE)
    ctx.set_material(mat)
    ctx.build_pattern()
    return ctx


def _ele(etype):
    from femcy_amd import element_zoo as ez
    return {"C3D4": ez.Element_linear_tetrahedral, "C3D10": ez.Element_quadratic_tetrahedral,
            "CPE8": ez.Element_quadratic_quadrilateral}[etype]()


@pytest.mark.parametrize("etype", ["C3D4", "C3D10"])
def test_one_element_Ke_equals_exact_integration(gpu_ctx_factory, etype):
    from types import SimpleNamespace
    from femcy_amd import backend as be
    import sympy_pins as spn
    Ke, X, C = spn.exact_Ke(etype)
    el = np.arange(X.shape[0], dtype=np.int32)[None, :]
    mat = SimpleNamespace(kind=be_kind("lin3d"), C=C, params=np.array([1.0, 0.25]))
    ctx = _ctx(gpu_ctx_factory, X, el, _ele(etype), mat)
    modes = [be.ASM_GATHER, be.ASM_GATHER_SYM, be.ASM_GATHER_SYM_ROWSUM, be.ASM_ROWS, be.ASM_ROWS2, be.ASM_ROWS3, be.ASM_ATOMIC, be.ASM_AUTO]
    if etype == "C3D10":
        modes.append(be.ASM_ROWS4)
    for mode in modes:
        ctx.set_option(be.OPT_ASSEMBLY, mode)
        ctx.assemble_K(-1)                                    # undeformed configuration
        K = ctx.get_K_bsr().toarray()
        assert np.abs(K - Ke).max() < 1e-13 * np.abs(Ke).max(), mode


@pytest.mark.parametrize("etype", ["C3D4", "C3D10"])
def test_unreferenced_nodes_and_two_elements_sharing_a_face(gpu_ctx_factory, etype):
    """edge cases of the row-centric assemblies: nodes no element refers to (rows without incident elements: the
    stored row is its zero diagonal block), a last slice with an odd number of rows, rows with one and with two
    incident elements -- two elements sharing a face, every assembly variant against the sum of the two exactly
    integrated element matrices"""
    from types import SimpleNamespace
    from femcy_amd import backend as be
    import sympy_pins as spn
    Ke, X, C = spn.exact_Ke(etype)
    npe = X.shape[0]
    # mirror image of the element through its face opposite to corner 0 (the plane of corners 1, 2, 3): the face's
    # nodes are shared, the others are new; corner / mid-side numbering of the mirrored element follows the original's
    p1, p2, p3 = X[1], X[2], X[3]
    nrm = np.cross(p2 - p1, p3 - p1)
    nrm /= np.linalg.norm(nrm)
    mirror = lambda P: P - 2.0 * np.outer((P - p1) @ nrm, nrm)
    on_face = np.abs((X - p1) @ nrm) < 1e-12
    Xm = mirror(X)
    new_ids = {}
    nodes = [*X]
    el2 = []
    for a in range(npe):
        if on_face[a]:
            el2.append(a)
        else:
            new_ids[a] = len(nodes)
            nodes.append(Xm[a])
            el2.append(new_ids[a])
    # a reflection flips the orientation: swap two corners (and the mid-side nodes that go with them) to keep det J > 0
    if etype == "C3D4":
        el2[1], el2[2] = el2[2], el2[1]
    else:
        el2[1], el2[2] = el2[2], el2[1]
        el2[4], el2[6] = el2[6], el2[4]        # mid-sides (0,1) <-> (0,2)
        el2[8], el2[9] = el2[9], el2[8]        # mid-sides (1,3) <-> (2,3)
    nodes += [np.array([9.0, 9.0, 9.0]), np.array([8.0, 9.0, 7.0]), np.array([7.0, 7.0, 9.0])]     # never referenced
    nodes = np.array(nodes)
    el = np.array([list(range(npe)), el2], dtype=np.int32)
    topo = orc.Topology(nodes, el, elem_def(etype))
    Ko = orc.assemble_K(topo, np.zeros(nodes.size), C)
    assert np.abs(Ko.toarray()[:3 * npe, :3 * npe] - Ke).max() < 2.0 * np.abs(Ke).max()      # sanity: same scale
    mat = SimpleNamespace(kind=be_kind("lin3d"), C=C, params=np.array([1.0, 0.25]))
    ctx = _ctx(gpu_ctx_factory, nodes, el, _ele(etype), mat)
    modes = [be.ASM_GATHER, be.ASM_GATHER_SYM, be.ASM_GATHER_SYM_ROWSUM, be.ASM_ROWS, be.ASM_ROWS2, be.ASM_ROWS3, be.ASM_ATOMIC, be.ASM_AUTO]
    if etype == "C3D10":
        modes.append(be.ASM_ROWS4)
    for mode in modes:
        ctx.set_option(be.OPT_ASSEMBLY, mode)
        ctx.assemble_K(-1)
        K = ctx.get_K_bsr().toarray()
        assert np.abs(K - Ko.toarray()).max() < 1e-12 * np.abs(Ke).max(), mode
        assert np.abs(K[-9:, :]).max() == 0.0 and np.abs(K[:, -9:]).max() == 0.0, mode


def be_kind(name):
    return {"lin3d": 0, "pstrain": 1, "pstress": 2, "neohooke": 3}[name]


@pytest.mark.parametrize("etype", ["C3D4", "C3D10"])
@pytest.mark.parametrize("material", ["stvk", "neohooke"])
def test_one_element_homogeneous_deformation(gpu_ctx_factory, etype, material):
    import sympy as sp
    from femcy_amd import backend as be
    from femcy_amd.material_zoo import LinearIsotropic, NeoHookean
    import sympy_pins as spn
    if material == "stvk":
        lam, mu = 1.5, 1.25
        spec = ("stvk", sp.Rational(3, 2), sp.Rational(5, 4))
        mat = LinearIsotropic(mu * (3 * lam + 2 * mu) / (lam + mu), lam / (2 * (lam + mu)))
    else:
        spec = ("neohooke", sp.Rational(2, 5), sp.Rational(1, 4))
        mat = NeoHookean(0.4, 0.25)
    f, u, X, F, sig = spn.homogeneous_case(etype, spec)
    el = np.arange(X.shape[0], dtype=np.int32)[None, :]
    ctx = _ctx(gpu_ctx_factory, X, el, _ele(etype), mat)
    ctx.upload(be.VEC_DOF, u)
    ctx.residual_and_K(be.VEC_DOF, be.VEC_FORCE)
    assert np.abs(ctx.download(be.VEC_FORCE) - f).max() < 1e-13 * np.abs(f).max()
    Fg = ctx.gauss_field(be.GP_F).to_numpy()
    Sg = ctx.gauss_field(be.GP_SIGMA).to_numpy()
    assert np.abs(Fg - F).max() < 1e-14 and np.abs(Sg - sig).max() < 1e-13 * np.abs(sig).max()


def _closed_form_cauchy(kind, F, params):
    """written out here, independent of oracle/ and of the kernels (material_zoo docstrings give the laws)"""
    dm = F.shape[0]
    J = np.linalg.det(F)
    if kind == "neohooke":
        C1, D1 = params
        return 2 * C1 / J * (F @ F.T - np.eye(dm)) + 2 * D1 * (J - 1) * np.eye(dm)
    E_, nu = params
    lam, mu = E_ * nu / ((1 + nu) * (1 - 2 * nu)), E_ / (2 * (1 + nu))
    E = (F.T @ F - np.eye(dm)) / 2
    S = lam * np.trace(E) * np.eye(dm) + 2 * mu * E              # 3-D St.Venant-Kirchhoff; plane strain: E33 = 0
    return F @ S @ F.T / J


@pytest.mark.parametrize("name,kind", [("twist_plate_C3D4.inp", "lin3d"), ("twist_C3D10_coarse.inp", "lin3d"),
                                       ("cook_3d_linearEl_largeDef.inp", "neohooke"), ("gen_beam_CPE8_tip4.inp", "pstrain")])
def test_affine_patch(gpu_ctx_factory, name, kind):
    """homogeneous F on a whole mesh: sigma = closed form at every Gauss point, interior nodes in equilibrium"""
    from femcy_amd import backend as be
    from femcy_amd.body import Body
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    et = list(inp.eSets)[0]
    el = inp.eSets[et]
    mat = list(inp.materials.values())[0]
    assert oracle_material(mat).kind == kind
    dm = inp.nodes.shape[1]
    F = np.array([[1.08, 0.05, -0.03], [0.02, 0.95, 0.06], [-0.04, 0.07, 1.04]])[:dm, :dm]
    u = (inp.nodes @ (F - np.eye(dm)).T).ravel()
    ctx = _ctx(gpu_ctx_factory, inp.nodes, el, inp.ELE, mat)
    ctx.upload(be.VEC_DOF, u)
    ctx.residual_and_K(be.VEC_DOF, be.VEC_FORCE)
    f = ctx.download(be.VEC_FORCE).reshape(-1, dm)
    sig = _closed_form_cauchy(kind, F, [float(v) for v in mat.params])
    Sg = ctx.gauss_field(be.GP_SIGMA).to_numpy()
    Fg = ctx.gauss_field(be.GP_F).to_numpy()
    assert np.abs(Fg - F).max() < 1e-12
    assert np.abs(Sg - sig).max() < 1e-11 * np.abs(sig).max()
    boundary_nodes = np.unique(np.concatenate([np.asarray(k) for k in Body(inp.nodes, el, inp.ELE).get_boundary().keys()]))
    interior = np.setdiff1d(np.arange(inp.nodes.shape[0]), boundary_nodes)
    assert interior.size > 0
    assert np.abs(f[interior]).max() < 1e-10 * np.abs(f).max()
    # total force = 0 and total moment = 0 (a homogeneous stress field is self-equilibrated)
    assert np.abs(f.sum(axis=0)).max() < 1e-10 * np.abs(f).max()


@pytest.mark.parametrize("name", ["twist_plate_C3D4.inp", "cook_3d_linearEl_largeDef.inp", "twist_C3D10_coarse.inp",
                                  "cookMembrane_2d_linearEl_largeDef.inp"])
def test_consistent_tangent_equals_oracle_complex_step(gpu_ctx_factory, name):
    from femcy_amd import backend as be
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    et = list(inp.eSets)[0]
    el = inp.eSets[et]
    mat = list(inp.materials.values())[0]
    topo = orc.Topology(inp.nodes, el, elem_def(et))
    L = np.ptp(inp.nodes, axis=0).max()
    x = inp.nodes / L
    u = (0.05 * L * np.stack([np.sin(1.3 * x[:, 0] + 0.4) * np.cos(0.7 * x[:, -1]), 0.5 * np.cos(2.1 * x[:, 1] - 0.2) * x[:, 0],
                              0.3 * np.sin(x.sum(axis=1))][:topo.dm], axis=1)).ravel()
    Ko = orc.consistent_tangent(topo, u, oracle_material(mat))
    ctx = _ctx(gpu_ctx_factory, inp.nodes, el, inp.ELE, mat)
    ctx.set_option(be.OPT_TANGENT, 1)
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)
    K = ctx.get_K_bsr().tocsr()
    assert abs(K - Ko).max() < 1e-10 * abs(Ko).max()
    # and the reference's matrix (B^T C B with the constant C) is NOT that derivative at 5 % strain
    ctx.set_option(be.OPT_TANGENT, 0)
    ctx.assemble_K(be.VEC_DOF)
    assert abs(ctx.get_K_bsr().tocsr() - Ko).max() > 1e-3 * abs(Ko).max()


# ------------------------------------------------------------------------------------------------ round 4: 2-D families
PLANE = ["CPS3", "CPS4", "CPS6", "CPS8"]


def _plane_ele(etype):
    from femcy_amd import element_zoo as ez
    return {"CPS3": ez.Element_linear_triangular, "CPS4": ez.Element_linear_quadrilateral,
            "CPS6": ez.Element_quadratic_triangular, "CPS8": ez.Element_quadratic_quadrilateral}[etype]()


def _plane_mat(mkind):
    from femcy_amd.material_zoo import LinearIsotropicPlaneStrain, LinearIsotropicPlaneStress
    return (LinearIsotropicPlaneStrain if mkind == "pstrain" else LinearIsotropicPlaneStress)(3.5, 0.3)


@pytest.mark.parametrize("etype", PLANE)
@pytest.mark.parametrize("mkind", ["pstrain", "pstress"])
def test_one_2d_element_Ke_equals_the_symbolic_matrix(gpu_ctx_factory, etype, mkind):
    """every assembly variant that exists for 2-D elements against tests/sympy_pins2d.py: the exactly integrated K^e
    (CPS3, CPS6, CPS4 on affine geometry), the reduced 2 x 2 matrix of the exact integrand for CPS8"""
    import sympy as sp
    from femcy_amd import backend as be
    import sympy_pins2d as sp2
    Ke, Kr, X, C = sp2.exact_Ke(sp2.ABAQUS[etype], (mkind, sp.Rational(7, 2), sp.Rational(3, 10)))
    el = np.arange(X.shape[0], dtype=np.int32)[None, :]
    ctx = _ctx(gpu_ctx_factory, X, el, _plane_ele(etype), _plane_mat(mkind))
    for mode in (be.ASM_GATHER, be.ASM_ATOMIC, be.ASM_ROWS, be.ASM_AUTO, be.ASM_GATHER_SYM, be.ASM_GATHER_SYM_ROWSUM, be.ASM_PAIRS):
        ctx.set_option(be.OPT_ASSEMBLY, mode)
        ctx.assemble_K(-1)
        K = ctx.get_K_bsr().toarray()
        assert np.abs(K - Kr).max() < 1e-13 * np.abs(Kr).max(), mode
    if etype != "CPS8":
        assert np.abs(K - Ke).max() < 1e-13 * np.abs(Ke).max()


@pytest.mark.parametrize("etype", PLANE)
def test_pair_list_assembly_edge_cases_and_general_C(gpu_ctx_factory, etype):
    """FEMCY_ASM_PAIRS (round 6) where its bookkeeping can go wrong: a generated mesh whose node count is not a multiple of
    16 or 64 (a last chunk / slice with padding rows), nodes no element refers to (rows without pairs: a zero diagonal
    block), chunks with more than 64 pairs (a fan of triangles around one node), a fully populated anisotropic C (the
    geometric-sum form is linear in C[v(i,j)][v(k,l)], not only in isotropic constants), a deformed configuration;
    against the oracle's scatter assembly, and bit-identical when repeated."""
    from types import SimpleNamespace
    from femcy_amd import backend as be
    ed = elem_def(etype)
    rng = np.random.default_rng(3)
    if etype in ("CPS3", "CPS6"):
        # a fan of triangles around node 0 (CPS3: one row with 40 incident elements, > 64 pairs in its chunk; CPS6: 16,
        # a 49-block row whose tile still fits the LDS), then a strip
        nfan = 40 if etype == "CPS3" else 16
        ang = np.linspace(0.0, 2 * np.pi, nfan, endpoint=False)
        pts = [np.zeros(2)] + [np.array([np.cos(a), np.sin(a)]) * (1.0 + 0.1 * np.sin(3 * a)) for a in ang]
        tris = [(0, 1 + k, 1 + (k + 1) % nfan) for k in range(nfan)]
        for k in range(23):                                      # the strip, attached to nothing
            b = len(pts)
            pts += [np.array([3.0 + k, 0.0]), np.array([4.0 + k, 0.1]), np.array([3.3 + k, 1.0])]
            tris.append((b, b + 1, b + 2))
        pts = np.array(pts)
        corners = np.array(tris, dtype=np.int32)
        quads = False
    else:
        nx, ny = 13, 5
        gx, gy = np.meshgrid(np.arange(nx + 1.0), np.arange(ny + 1.0), indexing="ij")
        pts = np.stack([gx.ravel() + 0.1 * np.sin(gy.ravel()), gy.ravel() + 0.07 * np.cos(gx.ravel())], axis=1)
        idx = lambda i, j: i * (ny + 1) + j
        corners = np.array([(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)) for i in range(nx) for j in range(ny)],
                           dtype=np.int32)
        quads = True
    if etype in ("CPS6", "CPS8"):                                 # mid-side nodes, one per distinct edge
        pts = [p for p in pts]
        mids = {}
        el = []
        nc = corners.shape[1]
        for c in corners:
            row = list(c)
            for k in range(nc):
                key = tuple(sorted((int(c[k]), int(c[(k + 1) % nc]))))
                if key not in mids:
                    mids[key] = len(pts)
                    pts.append(0.5 * (pts[key[0]] + pts[key[1]]) + 0.01 * rng.standard_normal(2))
                row.append(mids[key])
            el.append(row)
        pts = np.array(pts)
        el = np.array(el, dtype=np.int32)
    else:
        el = corners
    pts = np.vstack([pts, [[50.0, 50.0], [51.0, 50.0], [50.0, 51.0]]])     # never referenced
    assert ed.npe == el.shape[1]
    A = rng.standard_normal((3, 3))
    base = _plane_mat("pstrain")
    C = np.asarray(base.C) + 0.2 * np.abs(base.C).max() * (A + A.T)
    mat = SimpleNamespace(kind=base.kind, C=C, params=base.params)
    ctx = _ctx(gpu_ctx_factory, pts, el, _plane_ele(etype), mat)
    topo = orc.Topology(pts, el, ed)
    u = 0.02 * rng.standard_normal(pts.size)
    ctx.upload(be.VEC_DOF, u)
    Ko = orc.assemble_K(topo, u, C)
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_PAIRS)
    ctx.assemble_K(be.VEC_DOF)
    K1 = ctx.get_K_bsr()
    assert abs(K1.tocsr() - Ko).max() < 1e-12 * abs(Ko).max()
    assert np.abs(K1.toarray()[-6:, :]).max() == 0.0 and np.abs(K1.toarray()[:, -6:]).max() == 0.0
    ctx.assemble_K(be.VEC_DOF)
    K2 = ctx.get_K_bsr()
    assert np.array_equal(K1.data, K2.data) and np.array_equal(K1.indices, K2.indices)
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_GATHER)
    ctx.assemble_K(be.VEC_DOF)
    assert abs(ctx.get_K_bsr().tocsr() - K1.tocsr()).max() < 1e-12 * abs(Ko).max()


@pytest.mark.parametrize("etype", PLANE)
@pytest.mark.parametrize("mkind", ["pstrain", "pstress"])
def test_one_2d_element_homogeneous_deformation(gpu_ctx_factory, etype, mkind):
    """F, Cauchy stress (plane stress: the synthesised F33 of linear_isotropic_plane_stress.py:72-96) and nodal forces
    of one element under a homogeneous finite deformation against the closed form"""
    import sympy as sp
    from femcy_amd import backend as be
    import sympy_pins2d as sp2
    f, u, X, F, sig = sp2.homogeneous_case(sp2.ABAQUS[etype], (mkind, sp.Rational(7, 2), sp.Rational(3, 10)))
    el = np.arange(X.shape[0], dtype=np.int32)[None, :]
    ctx = _ctx(gpu_ctx_factory, X, el, _plane_ele(etype), _plane_mat(mkind))
    ctx.upload(be.VEC_DOF, u)
    ctx.residual_and_K(be.VEC_DOF, be.VEC_FORCE)
    assert np.abs(ctx.download(be.VEC_FORCE) - f).max() < 1e-13 * np.abs(f).max()
    Fg = ctx.gauss_field(be.GP_F).to_numpy()
    Sg = ctx.gauss_field(be.GP_SIGMA).to_numpy()
    assert np.abs(Fg - F).max() < 1e-14 and np.abs(Sg - sig).max() < 1e-13 * np.abs(sig).max()


@pytest.mark.parametrize("name,hyperelastic", [("twist_plate_C3D4.inp", True), ("twist_C3D10_coarse.inp", True),
                                               ("cook_3d_linearEl_largeDef.inp", True),
                                               ("cookMembrane_2d_linearEl_largeDef.inp", True),
                                               ("gen_beam_CPE8_tip4.inp", True),
                                               ("beamDeflec_quadPSE_largeD_load800.inp", False)])
def test_device_force_is_the_gradient_of_the_device_energy(gpu_ctx_factory, name, hyperelastic):
    """no oracle in this one: `femcy_internal_force` (sigma(F), current gradients and volumes, node gather) against the
    central difference of `femcy_elastic_energy` (F, energy density, reference volumes) along random directions -- two
    device paths that share only the deformation gradient.  Equal for the hyperelastic laws (St. Venant-Kirchhoff,
    neo-Hookean, plane strain), and NOT for the reference's plane-stress class, whose F33 does not come from its energy
    (tests/test_oracle_pins.py::test_internal_force_is_the_gradient_of_the_strain_energy says the same of the oracle)."""
    from femcy_amd import backend as be
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    el = list(inp.eSets.values())[0]
    ctx = _ctx(gpu_ctx_factory, inp.nodes, el, inp.ELE, list(inp.materials.values())[0])
    L = np.ptp(inp.nodes, axis=0).max()
    x = inp.nodes / L
    u = (0.03 * L * np.stack([np.sin(1.3 * x[:, 0] + 0.4) * np.cos(0.7 * x[:, -1]), 0.5 * np.cos(2.1 * x[:, 1] - 0.2) * x[:, 0],
                              0.3 * np.sin(x.sum(axis=1))][:inp.nodes.shape[1]], axis=1)).ravel()
    ctx.upload(be.VEC_DOF, u)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f = ctx.download(be.VEC_FORCE)
    ctx.assemble_K(-1)                                   # a geometry pass at u = 0: femcy_elastic_energy sums over the
    rng = np.random.default_rng(0)                       # volumes of the last pass (reference behaviour) = vol_0
    h = 1e-7 * L                                         # truncation ~ h^2: 6e-8 at worst here, rounding below that
    worst = 0.0
    for _ in range(3):
        v = rng.standard_normal(u.size)
        ctx.upload(be.VEC_TMP0, u + h * v)
        wp = ctx.elastic_energy(be.VEC_TMP0)
        ctx.upload(be.VEC_TMP0, u - h * v)
        wm = ctx.elastic_energy(be.VEC_TMP0)
        worst = max(worst, abs((wp - wm) / (2 * h) - f @ v) / abs(f @ v))
    print(f"{name}: |dW/du.v - f.v| / |f.v| = {worst:.2e}")
    if hyperelastic:
        assert worst < 1e-6
    else:
        assert 1e-3 < worst < 0.5
