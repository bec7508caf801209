"""-m gpu: femcy_direct_solve, the device's answer to the reference's direct branch (`solve_by_scipy`,
/root/reference/stiffnessMtrx.py:219-251: scipy `spsolve` below 1e5 DOF).  The checker is what the reference calls:
scipy's sparse LU on the matrix exported from the device (`femcy_get_K_bsr`), with seeded right-hand sides -- on
positive definite systems (every element family, nu -> 0.5), on the indefinite K of a configuration with inverted
elements (the reference's LU solves those too and the increment driver's path depends on the answer), on singular
systems (reported, never returned), and through the increment / Newton driver.  The same file runs against
libfemcy_cpu.so in the CPU suite (tests/test_cpu_backend.py)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from helpers import deck

pytestmark = pytest.mark.gpu


def load(name):
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    et = list(inp.eSets)[0]
    return inp, inp.eSets[et], list(inp.materials.values())[0]


def make_ctx(factory, inp, el, mat):
    ctx = factory()
    ctx.set_mesh(inp.nodes, el)
    ctx.set_element(inp.ELE)
    ctx.set_material(mat)
    ctx.build_pattern()
    return ctx


def constrained(inp, dm):
    return np.unique(np.concatenate([np.asarray(b["node_set"]) * dm + b["dof"] for b in inp.dirichlet_bc_info]))


@pytest.mark.parametrize("name,tol", [
    ("beam_CPS3_disp_meshSize1.inp", 1e-10),                       # tri3, one panel
    ("ellip_CPS4.inp", 1e-10), ("ellip_CPS8.inp", 1e-10),          # quad4 / quad8
    ("cookMembrane_CPE6_smallDef.inp", 1e-10),                     # tri6
    ("ellip_dense_CPS6_0d04.inp", 1e-10),                          # 29 k DOF: ~900 column panels
    ("cookMembrane_CPE6_smallDef_nu0d4999.inp", 1e-6),             # nu -> 0.5: condition ~1e8 (the tight PCG needs 4 n iterations)
    ("twist_plate_C3D4.inp", 1e-10), ("twist_C3D10_coarse.inp", 1e-10), ("cook_3d_quadEl_smallDef.inp", 1e-10),
])
def test_direct_solve_equals_sparse_lu(gpu_ctx_factory, name, tol):
    from femcy_amd import backend as be
    inp, el, mat = load(name)
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    dm = ctx.dm
    ctx.assemble_K(-1)
    rng = np.random.default_rng(7)
    b = rng.standard_normal(ctx.n)
    ctx.upload(be.VEC_RESIDUAL, b)
    ctx.dirichlet_newton(constrained(inp, dm), be.VEC_RESIDUAL)
    b = ctx.download(be.VEC_RESIDUAL)
    K = ctx.get_K_bsr().tocsc()
    info = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    x = ctx.download(be.VEC_X)
    ref = spl.spsolve(K, b)
    assert info["n"] == ctx.n and info["negative_pivots"] == 0 and info["singular_at"] == 0
    assert 0 < info["bandwidth"] < ctx.n
    assert np.array_equal(ctx.download(be.VEC_RESIDUAL), b)        # the right-hand side is left alone
    res = np.abs(K @ x - b).max() / np.abs(b).max()
    res_lu = np.abs(K @ ref - b).max() / np.abs(b).max()
    err = np.linalg.norm(x - ref) / np.linalg.norm(ref)
    print(f"{name}: n = {ctx.n}, bandwidth {info['bandwidth']}, panels {info['panels']}, |x - x_lu| / |x_lu| = {err:.2e}, "
          f"residual {res:.2e} (sparse LU: {res_lu:.2e}), refinements {info['refinements']}")
    # the residual a backward-stable solver leaves scales with |K| |x| / |b|: 1e-11 on the ordinary systems, the LU's own
    # level (x 10) on the nearly incompressible one
    assert max(res, info["residual"]) <= max(1e-11, 10 * res_lu), (res, res_lu, info)
    assert err <= tol
    # a second solve on the same context (storage reused), another right-hand side
    b2 = b[::-1].copy()
    b2[constrained(inp, dm)] = 0.0
    ctx.upload(be.VEC_TMP0, b2)
    ctx.direct_solve(be.VEC_TMP0, be.VEC_TMP1)
    x2 = ctx.download(be.VEC_TMP1)
    assert np.abs(K @ x2 - b2).max() / np.abs(b2).max() <= max(1e-11, 10 * res_lu)


@pytest.mark.parametrize("name", ["ellip_dense_CPS6_0d04.inp", "twist_plate_C3D10.inp", "cookMembrane_CPE6_smallDef.inp"])
def test_reverse_cuthill_mckee_makes_the_band_narrow(gpu_ctx_factory, name):
    """the decks' numberings couple nodes far apart; after the library's renumbering the band is a small multiple of the
    mesh's cross-section -- and no wider than what scipy's reverse Cuthill-McKee finds on the same graph (x 1.25)"""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    from femcy_amd import backend as be
    inp, el, mat = load(name)
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, np.ones(ctx.n))
    ctx.dirichlet_newton(constrained(inp, ctx.dm), be.VEC_RESIDUAL)
    info = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    natural = (int(np.abs(el[:, :, None] - el[:, None, :]).max()) + 1) * ctx.dm - 1
    npe = el.shape[1]
    G = sp.csr_matrix((np.ones(el.size * npe, dtype=np.int8), (np.repeat(el, npe, axis=1).ravel(), np.tile(el, (1, npe)).ravel())),
                      shape=(ctx.nn, ctx.nn))
    perm = reverse_cuthill_mckee(G, symmetric_mode=True)
    rank = np.empty(ctx.nn, dtype=np.int64)
    rank[perm] = np.arange(ctx.nn)
    scipy_bw = (int(np.abs(rank[el][:, :, None] - rank[el][:, None, :]).max()) + 1) * ctx.dm - 1
    print(f"{name}: sub-diagonals: caller's numbering {natural}, library {info['bandwidth']}, scipy's RCM {scipy_bw} (n = {ctx.n})")
    assert info["bandwidth"] < natural and info["bandwidth"] <= 1.25 * scipy_bw
    assert info["bandwidth"] < 0.2 * ctx.n


def test_indefinite_matrix_of_inverted_elements_is_still_solved(gpu_ctx_factory):
    """a Newton iterate on its way out: elements turned inside out, det J < 0 there, K = sum B^T C B det J w indefinite.
    The reference's LU returns the solution of that system and the driver iterates on; so does L S L^T + refinement."""
    from femcy_amd import backend as be
    inp, el, mat = load("twist_plate_C3D4.inp")
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    rng = np.random.default_rng(11)
    h = np.linalg.norm(inp.nodes[el[:, 0]] - inp.nodes[el[:, 1]], axis=1).mean()     # a typical edge
    u = np.zeros_like(inp.nodes)
    pick = rng.choice(inp.nodes.shape[0], inp.nodes.shape[0] // 12, replace=False)
    u[pick] = rng.standard_normal((pick.size, 3)) * 0.8 * h              # a few nodes pushed through their neighbours
    ctx.upload(be.VEC_DOF, u.ravel())
    ctx.assemble_K(be.VEC_DOF)
    vol = ctx.gauss_field(be.GP_VOL).to_numpy()
    assert (vol < 0).any() and (vol > 0).any()
    b = rng.standard_normal(ctx.n)
    ctx.upload(be.VEC_RESIDUAL, b)
    ctx.dirichlet_newton(constrained(inp, 3), be.VEC_RESIDUAL)
    b = ctx.download(be.VEC_RESIDUAL)
    K = ctx.get_K_bsr().tocsc()
    assert np.linalg.eigvalsh(K.toarray()).min() < 0.0
    info = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    x = ctx.download(be.VEC_X)
    ref = spl.spsolve(K, b)
    err = np.linalg.norm(x - ref) / np.linalg.norm(ref)
    print(f"{info['negative_pivots']} negative pivots, {info['refinements']} refinements, residual {info['residual']:.2e}, "
          f"|x - x_lu| / |x_lu| = {err:.2e}")
    assert info["negative_pivots"] > 0 and info["residual"] <= 1e-8
    assert err <= 1e-6


def test_negative_definite_matrix_all_signs_flipped(gpu_ctx_factory):
    """-K: every pivot negative, the same solution with the opposite sign (the signs S carry the whole difference)"""
    from femcy_amd import backend as be
    inp, el, mat = load("ellip_CPS4.inp")
    import copy
    neg = copy.copy(mat)
    neg.C = -np.asarray(mat.C.to_numpy() if hasattr(mat.C, "to_numpy") else mat.C)
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    ctx2 = make_ctx(gpu_ctx_factory, inp, el, neg)
    cons = constrained(inp, 2)
    free = np.setdiff1d(np.arange(ctx.n), cons)
    b = np.zeros(ctx.n)
    b[free] = np.random.default_rng(5).standard_normal(free.size)
    xs = []
    for c in (ctx, ctx2):
        c.assemble_K(-1)
        c.upload(be.VEC_RESIDUAL, b)
        c.dirichlet_newton(cons, be.VEC_RESIDUAL)
        info = c.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
        xs.append(c.download(be.VEC_X))
    assert info["negative_pivots"] == free.size                          # the unit rows of the constrained DOFs stay +1
    assert np.linalg.norm(xs[0] + xs[1]) <= 1e-12 * np.linalg.norm(xs[0])


def test_singular_and_nan_matrices_are_reported(gpu_ctx_factory):
    from femcy_amd import backend as be
    inp, el, mat = load("ellip_CPS4.inp")
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    ctx.assemble_K(-1)                                                   # no Dirichlet treatment: three rigid-body modes
    ctx.upload(be.VEC_RESIDUAL, np.random.default_rng(2).standard_normal(ctx.n))
    with pytest.raises(be.FemcyError) as e:
        ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    assert e.value.status == be.FEMCY_ENUMERIC
    u = np.zeros(ctx.n)
    u[5] = np.nan
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)                                           # NaN coordinates -> NaN blocks
    ctx.dirichlet_newton(constrained(inp, 2), be.VEC_RESIDUAL)
    with pytest.raises(be.FemcyError) as e:
        ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    assert e.value.status == be.FEMCY_ENUMERIC
    # and the context recovers
    ctx.assemble_K(-1)
    ctx.dirichlet_newton(constrained(inp, 2), be.VEC_RESIDUAL)
    assert ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)["residual"] <= 1e-9


def test_argument_and_resource_errors(gpu_ctx_factory):
    from femcy_amd import backend as be
    inp, el, mat = load("ellip_dense_CPS6_0d04.inp")
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, np.ones(ctx.n))
    ctx.dirichlet_newton(constrained(inp, 2), be.VEC_RESIDUAL)
    with pytest.raises(be.FemcyError):
        ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_RESIDUAL)               # in place
    with pytest.raises(be.FemcyError):
        ctx.set_option(be.OPT_DIRECT_MAX_BYTES, 1000)
    ctx.set_option(be.OPT_DIRECT_MAX_BYTES, 1 << 20)                     # this band takes ~140 MB
    with pytest.raises(be.FemcyError) as e:
        ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    assert e.value.status == be.FEMCY_ENOMEM and "FEMCY_OPT_DIRECT_MAX_BYTES" in str(e.value)
    ctx.set_option(be.OPT_DIRECT_MAX_BYTES, 1 << 30)
    assert ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)["residual"] <= 1e-9


def test_driver_takes_the_direct_branch_and_falls_back_loudly(capsys):
    """solve_dof below 1e5 DOF = the factorisation (no CG iteration at all); a band over the limit switches the system
    to the tight PCG with a printed notice, and the answers agree"""
    from femcy_amd import backend as be
    from femcy_amd.body import Body
    from femcy_amd.reader import InpInfo
    from femcy_amd.stiffnessMtrx import System_of_equations
    out = []
    for limit in (None, 1 << 20):
        inp = InpInfo(deck("cookMembrane_CPE6_smallDef_nu0d4999.inp"))
        body = Body(nodes=inp.nodes, elements=list(inp.eSets.values())[0], ELE=inp.ELE)
        system = System_of_equations(body, list(inp.materials.values())[0], inp.geometric_nonlinear, verbose=False)
        if limit:
            system.ctx.set_option(be.OPT_DIRECT_MAX_BYTES, limit)
        system.solve(inp)
        out.append((system.dof.to_numpy(), dict(system.stats), system.direct))
        system.ctx.close()
    (u0, s0, d0), (u1, s1, d1) = out
    assert d0 == "auto" and s0["direct_solves"] == s0["linear_solves"] == 1 and s0["cg_iterations"] == 0   # narrow band: the factorisation
    assert d1 == "pcg" and s1["direct_solves"] == 0 and s1["cg_iterations"] > 3498     # more than n iterations at nu = 0.4999
    assert "tight PCG instead" in capsys.readouterr().out
    assert np.linalg.norm(u1 - u0) <= 1e-6 * np.linalg.norm(u0)


def test_auto_chooses_between_the_factorisation_and_the_tight_pcg():
    """direct = "auto" (the default): the reference switches solvers on a DOF count alone (stiffnessMtrx.py:272-276); here
    the stand-in for its spsolve is chosen by the band first (femcy_direct_plan) and by measured times afterwards.
    A cube-like 3-D mesh (89 k DOF, 2 888 sub-diagonals: n * bw^2 = 7.5e11 flops behind 2 793 dependent panels) starts
    with the tight PCG; the 2-D decks and the slender twist plates start -- and stay -- with the factorisation.  Either
    way the answer is the factorisation's to 1e-9."""
    from types import SimpleNamespace
    from femcy_amd import meshgen
    from femcy_amd.body import Body
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from femcy_amd.stiffnessMtrx import System_of_equations
    m = meshgen.twist_plate(30, 30, 30)
    ELE = Element_linear_tetrahedral()
    ti = dict(m["time_incs"], ini_inc=0.002, max_inc=0.002, max_time=0.008)         # four small increments of twist: >= 4 solves
    inp = SimpleNamespace(nodes=m["nodes"], eSets={"C3D4": m["elements"]}, ELE=ELE, dirichlet_bc_info=m["dirichlet_bc_info"],
                          neumann_bc_info=[], time_incs=ti, geometric_nonlinear=True,
                          materials={"Elastic": LinearIsotropic(*m["elastic"])})
    res = {}
    for mode in ("auto", "auto-timed", "cholesky"):
        s = System_of_equations(Body(inp.nodes, m["elements"], ELE), inp.materials["Elastic"], True, verbose=False, direct=mode)
        plan = s.ctx.direct_plan()
        assert plan["n"] == m["nodes"].size and plan["bandwidth"] > s.AUTO_WIDE_BAND and plan["panels"] in (0, (plan["n"] + 31) // 32)   # (0: the host backend has no panels)
        s.solve(inp)
        res[mode] = (s.dof.to_numpy(), dict(s.stats), dict(s._auto), [dict(i) for i in s.increments], list(s.direct_log))
        s.ctx.close()
    (ua, sa, auto, inca, loga), (ut, st, timed, inct, logt), (uc, sc, _, incc, logc) = res["auto"], res["auto-timed"], res["cholesky"]
    # "auto" (default, round 6): the band decides once -- the same solver on every solve, on every run
    assert auto["first"] == auto["pick"] == "pcg" and not auto["ms"] and sa["cg_iterations"] > 0
    assert loga == ["pcg"] * sa["linear_solves"] and sa["linear_solves"] == sc["linear_solves"] >= 4
    assert logc == ["cholesky"] * sc["linear_solves"] and sc["cg_iterations"] == 0 and sc["direct_solves"] == sc["linear_solves"]
    # "auto-timed": a first solve above AUTO_TRY_OTHER_MS starts the exploration -- both methods timed twice, alternating
    # (the first call of either carries one-off costs), the faster one kept; otherwise the first choice stays
    assert timed["first"] == "pcg" and len(logt) == st["linear_solves"] == sc["linear_solves"]
    if timed["ms"]["pcg"] > System_of_equations.AUTO_TRY_OTHER_MS:
        assert logt[:4] == ["pcg", "cholesky", "pcg", "cholesky"] and timed["samples"]["pcg"] >= 2 and timed["samples"]["cholesky"] >= 2
        assert set(timed["ms"]) == {"pcg", "cholesky"} and timed["pick"] == min(timed["ms"], key=timed["ms"].get)
        assert all(v == timed["pick"] for v in logt[4:])
    else:
        assert timed["pick"] == "pcg" and logt == ["pcg"] * len(logt)
    for inc in (inca, inct):
        assert [(i["time1"], i["converged"], i["newton_loop"]) for i in inc] == [(i["time1"], i["converged"], i["newton_loop"]) for i in incc]
    assert np.linalg.norm(ua - uc) <= 1e-9 * np.linalg.norm(uc) and np.linalg.norm(ut - uc) <= 1e-9 * np.linalg.norm(uc)
    print(f"[auto] 30^3 cube: plan {plan}, auto {loga}, auto-timed {logt} ms {timed['ms']} pick {timed['pick']}")
    # a deck with a narrow band: the factorisation from the first solve on, the PCG never runs
    inp2, el2, mat2 = load("twist_plate_C3D4.inp")
    s = System_of_equations(Body(inp2.nodes, el2, inp2.ELE), mat2, inp2.geometric_nonlinear, verbose=False)
    assert s.direct == "auto" and s.ctx.direct_plan()["bandwidth"] < s.AUTO_WIDE_BAND
    inp2.time_incs = dict(inp2.time_incs, max_time=0.05)                 # the first increment of the deck
    s.solve(inp2)
    assert s._auto["first"] == s._auto["pick"] == "cholesky" and s.stats["linear_solves"] > 0
    assert s.stats["cg_iterations"] == 0 and s.stats["direct_solves"] == s.stats["linear_solves"]
    assert s.direct_log == ["cholesky"] * s.stats["linear_solves"]
    s.ctx.close()


@pytest.mark.parametrize("name,cube", [("twist_plate_C3D10.inp", 0), ("ellip_dense_CPS6_0d04.inp", 0), (None, 12), (None, 15)])
def test_tile_update_on_the_matrix_cores_equals_the_valu_product(gpu_ctx_factory, name, cube):
    """FEMCY_TUNE_DIRECT_UPDATE: the trailing update of the band factorisation as a VALU tile product (0), on the f64 matrix
    cores with one tile pair per workgroup (1), with 2 x 2 tile pairs per workgroup (2; an odd tile count leaves
    half-empty blocks on the last block row / the diagonal: 15^3 cells) and as 1 on two streams (3: the column the next
    panel needs first, the rest beside that panel) -- the same factor: solutions agree to rounding,
    each solves K x = b, indefinite K included"""
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    if name:
        inp, el, mat = load(name)
        ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
        cons = constrained(inp, ctx.dm)
    else:
        m = meshgen.twist_plate(cube, cube, cube)
        ctx = gpu_ctx_factory()
        ctx.set_mesh(m["nodes"], m["elements"])
        ctx.set_element(Element_linear_tetrahedral())
        ctx.set_material(LinearIsotropic(*m["elastic"]))
        ctx.build_pattern()
        cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    rng = np.random.default_rng(5)
    for indefinite in (False, True):
        if indefinite:                       # a displaced configuration with inverted elements: K = L S L^T with negative pivots
            u = 0.3 * np.ptp(ctx_nodes(ctx, name, cube), axis=0).max() * rng.standard_normal(ctx.n) / 20.0
            ctx.upload(be.VEC_DOF, u)
            ctx.assemble_K(be.VEC_DOF)
        else:
            ctx.assemble_K(-1)
        ctx.upload(be.VEC_RESIDUAL, rng.standard_normal(ctx.n))
        ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
        b = ctx.download(be.VEC_RESIDUAL)
        K = ctx.get_K_bsr().tocsr()
        xs, infos = {}, {}
        for var in (0, 1, 2, 3):
            ctx.set_option(be.TUNE_DIRECT_UPDATE, var)
            try:
                infos[var] = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
                xs[var] = ctx.download(be.VEC_X)
            except be.FemcyError as e:       # elimination without pivoting may lose an indefinite matrix: then every variant must
                assert e.status == be.FEMCY_ENUMERIC and indefinite
                infos[var] = None
        ctx.set_option(be.TUNE_DIRECT_UPDATE, -1)
        assert len({v is None for v in infos.values()}) == 1
        if infos[0] is None:
            continue
        assert infos[0]["bandwidth"] // 32 >= 8                     # wide enough for the matrix-core path to be the one in use
        assert infos[1]["negative_pivots"] == infos[2]["negative_pivots"] == infos[3]["negative_pivots"] == infos[0]["negative_pivots"]
        if indefinite:
            assert infos[0]["negative_pivots"] > 0
        assert np.array_equal(xs[3], xs[1])                         # the two-stream schedule runs the same kernels on the same data
        for var in (0, 1, 2, 3):
            assert np.abs(K @ xs[var] - b).max() <= 1e-8 * np.abs(b).max()
            assert np.linalg.norm(xs[var] - xs[0]) <= 1e-9 * np.linalg.norm(xs[0])
    if not name:                             # (a context of gpu_ctx_factory is closed by the fixture)
        ctx.close()


def ctx_nodes(ctx, name, cube):
    from femcy_amd import meshgen
    return load(name)[0].nodes if name else meshgen.twist_plate(cube, cube, cube)["nodes"]


def test_direct_solve_refuses_a_partitioned_system(gpu_ctx_factory):
    from femcy_amd import backend as be
    inp, el, mat = load("twist_plate_C3D4.inp")
    ctx = make_ctx(gpu_ctx_factory, inp, el, mat)
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, np.ones(ctx.n))
    ctx.dirichlet_newton(constrained(inp, 3), be.VEC_RESIDUAL)
    iface = np.arange(0, ctx.n, 7, dtype=np.int32)
    ctx.comm_init(0, 1, be.Context.comm_unique_id(), iface, np.arange(iface.size, dtype=np.int32), iface.size,
                  np.ones(ctx.n, dtype=np.uint8))
    with pytest.raises(be.FemcyError) as e:
        ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
    assert e.value.status == -5                                          # FEMCY_ECOMM


def test_band_order_follows_a_new_pattern_on_the_same_context(gpu_ctx_factory):
    """the renumbering and the band storage belong to a pattern: another mesh on the same context gets its own"""
    from femcy_amd import backend as be
    ctx = gpu_ctx_factory()
    seen = []
    for name in ("ellip_CPS8.inp", "cookMembrane_CPE6_smallDef.inp", "ellip_CPS8.inp"):
        inp, el, mat = load(name)
        ctx.set_mesh(inp.nodes, el)
        ctx.set_element(inp.ELE)
        ctx.set_material(mat)
        ctx.build_pattern()
        ctx.assemble_K(-1)
        b = np.random.default_rng(3).standard_normal(ctx.n)
        ctx.upload(be.VEC_RESIDUAL, b)
        ctx.dirichlet_newton(constrained(inp, 2), be.VEC_RESIDUAL)
        b = ctx.download(be.VEC_RESIDUAL)
        info = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
        x = ctx.download(be.VEC_X)
        K = ctx.get_K_bsr().tocsc()
        assert info["n"] == ctx.n == b.size
        assert np.abs(K @ x - b).max() <= 1e-11 * np.abs(b).max()
        seen.append((info["n"], info["bandwidth"]))
    assert seen[0] == seen[2] and seen[0] != seen[1]


def test_auto_survives_a_pcg_breakdown():
    """direct = "auto" on a wide band starts with the tight PCG.  On the indefinite K of a diverging Newton iterate CG breaks
    down (femcy_pcg: FEMCY_ENUMERIC on a NaN / Inf residual) -- the reference's spsolve returns a solution there and
    advance_inc's path depends on it, so that solve, and every later one, must go to the L S L^T factorisation instead of
    leaving solve_by_scipy (round-5 advisor finding): forced here by a band limit of 0 and a pcg that raises once."""
    from femcy_amd import backend as be
    from femcy_amd.body import Body
    from femcy_amd.stiffnessMtrx import System_of_equations
    out = {}
    for mode in ("cholesky", "auto"):
        inp, el, mat = load("twist_plate_C3D4.inp")
        inp.time_incs = dict(inp.time_incs, max_time=0.05)
        s = System_of_equations(Body(inp.nodes, el, inp.ELE), mat, inp.geometric_nonlinear, verbose=False, direct=mode)
        calls = {"n": 0}
        if mode == "auto":
            s.AUTO_WIDE_BAND = 0                                  # every band is "wide": the first solve goes to the PCG
            real = s.ctx.pcg

            def breaking_pcg(*args, **kw):
                calls["n"] += 1
                raise be.FemcyError("NaN residual after 1 iterations (injected)", status=be.FEMCY_ENUMERIC)

            s.ctx.pcg = breaking_pcg
        s.solve(inp)
        out[mode] = (s.dof.to_numpy(), dict(s.stats), list(s.direct_log), dict(s._auto), calls["n"],
                     [(i["time1"], i["converged"], i["newton_loop"]) for i in s.increments])
        if mode == "auto":
            s.ctx.pcg = real
        s.ctx.close()
    (uc, sc, logc, _, _, incc), (ua, sa, loga, auto, ncalls, inca) = out["cholesky"], out["auto"]
    assert ncalls == 1 and auto["first"] == "pcg" and auto["pcg_ok"] is False and auto["pick"] == "cholesky"
    assert loga == logc == ["cholesky"] * sc["linear_solves"] and sa["linear_solves"] == sc["linear_solves"] and inca == incc
    assert np.array_equal(ua, uc)                                 # the broken solve left dof untouched: the same bits
