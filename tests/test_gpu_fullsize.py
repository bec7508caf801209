"""-m gpu: BASELINE.json's full-size configuration (995 328 C3D4, 548 535 DOF, configs[2]) --
the HIP path against the as-written C/OpenMP oracle (which handles 1 M elements in seconds) and through
size-independent properties (symmetry, rigid-body null space, idempotence / run-to-run bit
reproducibility, Dirichlet identity rows, solver residual)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from femcy_amd.user_defined import user_dirichletBC_values
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    from oracle.femcy_oracle import Material
    m = meshgen.twist_plate_k(12)
    assert m["elements"].shape == (995328, 4) and m["nodes"].shape == (182845, 3)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern()
    assert (info.n, info.nnzb, info.nnz, info.max_row_blocks) == (548535, 2606005, 23454045, 15)
    u = np.zeros(ctx.n)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
        if bc["user"]:
            user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
    cons = np.unique(np.concatenate(cons))
    ed = elem_def("C3D4")
    co = COracle(m["nodes"], m["elements"], ed.dN_table(), ed.gauss_weights, Material("lin3d", m["elastic"]).C)
    yield dict(be=be, ctx=ctx, m=m, u=u, cons=cons, co=co)
    ctx.close()


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_fullsize_against_c_oracle(full):
    be, ctx, u, cons, co, m = full["be"], full["ctx"], full["u"], full["cons"], full["co"], full["m"]
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)
    co.get_dsdx_and_vol(u)
    co.assemble()
    assert rel(ctx.gauss_field(be.GP_VOL).to_numpy(), co.vol) < 1e-12
    assert rel(ctx.gauss_field(be.GP_DSDX).to_numpy(), co.dsdx) < 1e-12
    x = np.random.default_rng(0).standard_normal(ctx.n)
    ctx.upload(be.VEC_TMP0, x)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    y = ctx.download(be.VEC_TMP1)
    assert rel(y, co.compute_Ad(x)) < 1e-12                      # K(u) x, reference ELL SpMV as written
    # internal force (F at the reference configuration, StVK stress, current-configuration gather)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f = co.internal_force(u, 0, *m["elastic"])
    assert rel(ctx.download(be.VEC_FORCE), f) < 1e-11
    assert rel(ctx.gauss_field(be.GP_SIGMA).to_numpy(), co.sigma) < 1e-11
    # Newton Dirichlet + 30 iterations of the reference recurrence
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.assemble_K(be.VEC_DOF)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    co.get_dsdx_and_vol(u)
    co.assemble()
    co.zero_rows_cols_unit_diag(cons)
    f[cons] = 0.0
    assert rel(ctx.download(be.VEC_RESIDUAL), f) < 1e-11
    xo, ito, r0o, rmaxo = co.cg(f, eps=0.0, maxit=30)
    # both forms of the recurrence at this size: the persistent one-launch solve (default here) and three launches
    # per iteration (what a rank of a multi-GPU run executes)
    xs = {}
    for persist in (1, 0):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        before = ctx.timing()
        it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        after = ctx.timing()
        key = "solves_persist" if persist else "solves_three"
        assert after[key] - before[key] == 1
        assert it == ito == 30 and abs(r0 - r0o) < 1e-11 * r0o and abs(rmax - rmaxo) < 1e-6 * rmaxo
        xs[persist] = ctx.download(be.VEC_X)
        assert np.linalg.norm(xs[persist] - xo) / np.linalg.norm(xo) < 1e-8
    # (the two forms sum their reductions in different orders -- one in registers per wave, one over storage-order
    # partials: 1.2e-10 after 30 iterations on this matrix)
    assert np.linalg.norm(xs[1] - xs[0]) / np.linalg.norm(xs[0]) < 1e-9
    ctx.set_option(be.OPT_PCG_PERSIST, 1)


def test_fullsize_properties(full):
    be, ctx, u, cons = full["be"], full["ctx"], full["u"], full["cons"]
    rng = np.random.default_rng(1)
    ctx.upload(be.VEC_DOF, u)

    def Kx(v):
        ctx.upload(be.VEC_TMP0, v)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        return ctx.download(be.VEC_TMP1)

    ctx.assemble_K(be.VEC_DOF)
    x, y = rng.standard_normal(ctx.n), rng.standard_normal(ctx.n)
    Kx1, Ky1 = Kx(x), Kx(y)
    scale = np.abs(Kx1).max()
    assert abs(y @ Kx1 - x @ Ky1) < 1e-10 * abs(y @ Kx1)                 # symmetry
    assert np.abs(Kx(2.5 * x - 0.5 * y) - (2.5 * Kx1 - 0.5 * Ky1)).max() < 1e-12 * scale     # linearity
    for i in range(3):                                                   # rigid translations: K t = 0
        t = np.zeros(ctx.n)
        t[i::3] = 1.0
        assert np.abs(Kx(t)).max() < 1e-9 * scale
    # idempotence / determinism: re-assembling gives the same bits; the atomic variant the same values
    ctx.assemble_K(be.VEC_DOF)
    assert np.array_equal(Kx(x), Kx1)
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_ATOMIC)
    ctx.assemble_K(be.VEC_DOF)
    assert np.abs(Kx(x) - Kx1).max() < 1e-12 * scale
    for rows_mode in (be.ASM_ROWS, be.ASM_ROWS2, be.ASM_ROWS3):
        ctx.set_option(be.OPT_ASSEMBLY, rows_mode)
        ctx.assemble_K(be.VEC_DOF)
        y_rows = Kx(x)
        assert np.abs(y_rows - Kx1).max() < 1e-12 * scale
        ctx.assemble_K(be.VEC_DOF)
        assert np.array_equal(Kx(x), y_rows)                            # the row-centric variants are deterministic too
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_AUTO)
    ctx.assemble_K(be.VEC_DOF)
    # Dirichlet: constrained rows/columns become identity
    b = rng.standard_normal(ctx.n)
    ctx.upload(be.VEC_RESIDUAL, b)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    bb = ctx.download(be.VEC_RESIDUAL)
    assert (bb[cons] == 0).all() and np.array_equal(np.delete(bb, cons), np.delete(b, cons))
    Kxd = Kx(x)
    assert np.array_equal(Kxd[cons], x[cons])
    x0 = x.copy()
    x0[cons] = 0.0
    free = np.ones(ctx.n, bool)
    free[cons] = False
    assert np.abs(Kx(x0)[free] - Kxd[free]).max() < 1e-12 * scale        # columns of constrained DOFs are zero
    # the reference's CG settings at >= 1e5 DOF (eps = 1e-3, stiffnessMtrx.py:257-259) converge and solve
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
    xs = ctx.download(be.VEC_X)
    assert 0 < it < ctx.n and rmax < 1e-3 * r0
    assert np.abs(Kx(xs) - bb).max() < 1.0001e-3 * r0
    it2, _, _ = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
    assert it2 == it and np.array_equal(ctx.download(be.VEC_X), xs)       # run-to-run bit reproducible


def test_fullsize_force_is_the_gradient_of_the_energy(full):
    """a size-independent property with no oracle in it: at the bench's state S1 (5 % of the twist, finite rotations)
    the device's internal force (sigma(F), current gradients and volumes, node gather) is the derivative of the device's
    strain energy (F, energy density, reference volumes) -- central differences along two random directions"""
    be, ctx, u, m = full["be"], full["ctx"], full["u"], full["m"]
    ctx.upload(be.VEC_DOF, u)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f = ctx.download(be.VEC_FORCE)
    ctx.assemble_K(-1)                                   # geometry at u = 0: femcy_elastic_energy sums over vol_0
    rng = np.random.default_rng(9)
    h = 1e-7 * np.ptp(m["nodes"], axis=0).max()
    for _ in range(2):
        v = rng.standard_normal(ctx.n)
        ctx.upload(be.VEC_TMP0, u + h * v)
        wp = ctx.elastic_energy(be.VEC_TMP0)
        ctx.upload(be.VEC_TMP0, u - h * v)
        wm = ctx.elastic_energy(be.VEC_TMP0)
        err = abs((wp - wm) / (2 * h) - f @ v) / abs(f @ v)
        print(f"1 M C3D4: |dW/du.v - f.v| / |f.v| = {err:.2e}")
        assert err < 1e-5


# ----------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: the ~8 M-element C3D4 plate (k = 24: 7 962 624 elements, 4 183 275 DOF).  (a) on one GPU
# against the C oracle -- the matrix (1.55 GB) streams from HBM here, so this is the non-temporal / in-kernel-loop
# branch of the SpMV at the size it was written for; (b) the same mesh cut into 8 z-slabs, one context per slab
# joined by the in-process transport (the kernels and call sequence of the 8-GPU RCCL run), both interface
# exchanges, against the single-context result.
@pytest.fixture(scope="module")
def full8():
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from femcy_amd.user_defined import user_dirichletBC_values
    m = meshgen.twist_plate_k(24)
    assert m["elements"].shape == (7962624, 4) and m["nodes"].shape == (1394425, 3)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern()
    assert (info.n, info.nnzb, info.max_row_blocks) == (4183275, 20375785, 15)
    u = np.zeros(ctx.n)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
        if bc["user"]:
            user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
    cons = np.unique(np.concatenate(cons))
    yield dict(be=be, ctx=ctx, m=m, u=u, cons=cons, info=info)
    ctx.close()


def test_8M_single_gpu_against_c_oracle(full8):
    from helpers import node_adjacency
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    from oracle.femcy_oracle import Material
    be, ctx, u, cons, m = full8["be"], full8["ctx"], full8["u"], full8["cons"], full8["m"]
    ed = elem_def("C3D4")
    ptr, idx = node_adjacency(m["elements"], m["nodes"].shape[0])
    assert idx.size == full8["info"].nnzb                          # the device pattern = the reference's adjacency
    co = COracle(m["nodes"], m["elements"], ed.dN_table(), ed.gauss_weights, Material("lin3d", m["elastic"]).C, ptr, idx)
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)
    co.get_dsdx_and_vol(u)
    co.assemble()
    assert rel(ctx.gauss_field(be.GP_VOL).to_numpy(), co.vol) < 1e-12
    assert rel(ctx.gauss_field(be.GP_DSDX).to_numpy(), co.dsdx) < 1e-12
    x = np.random.default_rng(0).standard_normal(ctx.n)
    ctx.upload(be.VEC_TMP0, x)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    y = ctx.download(be.VEC_TMP1)
    assert rel(y, co.compute_Ad(x)) < 1e-12
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f = co.internal_force(u, 0, *m["elastic"])
    assert rel(ctx.download(be.VEC_FORCE), f) < 1e-11
    assert rel(ctx.gauss_field(be.GP_SIGMA).to_numpy(), co.sigma) < 1e-11
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.assemble_K(be.VEC_DOF)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    co.get_dsdx_and_vol(u)
    co.assemble()
    co.zero_rows_cols_unit_diag(cons)
    f[cons] = 0.0
    assert rel(ctx.download(be.VEC_RESIDUAL), f) < 1e-9
    # both sides iterate on the SAME right-hand side (the device's): the recurrence is what is compared
    b = ctx.download(be.VEC_RESIDUAL)
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
    xo, ito, r0o, rmaxo = co.cg(b, eps=0.0, maxit=30)
    assert it == ito == 30 and abs(r0 - r0o) < 1e-11 * r0o and abs(rmax - rmaxo) < 1e-6 * rmaxo
    assert np.linalg.norm(ctx.download(be.VEC_X) - xo) / np.linalg.norm(xo) < 1e-8
    # size-independent properties at this size
    rng = np.random.default_rng(2)

    def Kx(v):
        ctx.upload(be.VEC_TMP0, v)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        return ctx.download(be.VEC_TMP1)

    ctx.assemble_K(be.VEC_DOF)
    z = rng.standard_normal(ctx.n)
    Kx1, Kz1 = Kx(x), Kx(z)
    scale = np.abs(Kx1).max()
    assert abs(z @ Kx1 - x @ Kz1) < 1e-10 * abs(z @ Kx1)                                   # symmetry
    assert np.abs(Kx(2.5 * x - 0.5 * z) - (2.5 * Kx1 - 0.5 * Kz1)).max() < 1e-12 * scale   # linearity
    t = np.zeros(ctx.n)
    t[1::3] = 1.0
    assert np.abs(Kx(t)).max() < 1e-9 * scale                                              # rigid translation
    ctx.assemble_K(be.VEC_DOF)
    assert np.array_equal(Kx(x), Kx1)                                                      # bit-reproducible assembly
    ctx.upload(be.VEC_RESIDUAL, f)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    it1, r01, rm1 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-2, maxit=400)
    xs = ctx.download(be.VEC_X)
    it2, _, rm2 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-2, maxit=400)
    assert (it2, rm2) == (it1, rm1) and np.array_equal(ctx.download(be.VEC_X), xs)         # and solve
    full8["f_dirichlet"] = f


def test_8M_eight_slab_partition_equals_single_context(full8):
    from femcy_amd import meshgen, partition
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from test_gpu_multirank import run_ranks
    be, ctx, u_g, cons_g, m = full8["be"], full8["ctx"], full8["u"], full8["cons"], full8["m"]
    nx, ny, nz = m["cells"]
    nranks = 8
    rng = np.random.default_rng(7)
    x_g = rng.standard_normal(ctx.n)
    # single-context reference on the whole mesh
    ctx.upload(be.VEC_DOF, u_g)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f_ref = ctx.download(be.VEC_FORCE)
    ctx.assemble_K(be.VEC_DOF)
    ctx.upload(be.VEC_TMP0, x_g)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    y_ref = ctx.download(be.VEC_TMP1)
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.dirichlet_newton(cons_g, be.VEC_RESIDUAL)
    hist_ref = []
    for maxit in (1, 25):
        res = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=maxit)
        hist_ref.append((res, ctx.download(be.VEC_X)))
    uid = be.Context.comm_local_id()
    plane_dofs = (nx + 1) * (ny + 1) * 3

    def rank_main(r):
        p = partition.plate_slab_part(nx, ny, nz, nranks, r)      # the rank's own slab only, as bench.py builds it
        assert p.elements.shape[0] == 995328 and p.niface_global == 7 * plane_dofs
        c = be.Context(0)
        try:
            c.set_mesh(p.nodes, p.elements)
            c.set_element(Element_linear_tetrahedral())
            c.set_material(LinearIsotropic(*m["elastic"]))
            c.build_pattern()
            c.comm_init(r, nranks, uid, p.iface_local_dofs, p.iface_global_slot, p.niface_global, p.owner)
            c.comm_set_neighbours(p)
            bcs, _ = meshgen.twist_plate_bcs(p.nodes)
            cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in bcs])).astype(np.int32)
            out = {}
            for exch in (0, 1):
                c.set_option(be.OPT_EXCHANGE, exch)
                c.upload(be.VEC_DOF, p.scatter_global(u_g))
                c.internal_force(be.VEC_DOF, be.VEC_FORCE)
                f = c.download(be.VEC_FORCE)
                c.assemble_K(be.VEC_DOF)
                c.upload(be.VEC_TMP0, p.scatter_global(x_g))
                c.spmv(be.VEC_TMP0, be.VEC_TMP1)
                y = c.download(be.VEC_TMP1)
                c.vector(be.VEC_RHS).fill(0.0)
                c.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
                c.dirichlet_newton(cons, be.VEC_RESIDUAL)
                hist = []
                for maxit in (1, 25):
                    res = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=maxit)
                    hist.append((res, c.download(be.VEC_X)))
                out[exch] = (f, y, hist)
            return p, out
        finally:
            c.close()

    outs = run_ranks(nranks, rank_main)
    sf, sy = np.abs(f_ref).max(), np.abs(y_ref).max()
    for p, out in outs:
        for exch in (0, 1):
            f, y, hist = out[exch]
            assert np.abs(f - p.scatter_global(f_ref)).max() < 1e-11 * sf
            assert np.abs(y - p.scatter_global(y_ref)).max() < 1e-12 * sy
            for ((k, r0k, rmaxk), xk), ((kr, r0r, rmaxr), xr) in zip(hist, hist_ref):
                assert k == kr and abs(r0k - r0r) <= 1e-12 * r0r and abs(rmaxk - rmaxr) <= 1e-8 * rmaxr
                assert np.linalg.norm(xk - p.scatter_global(xr)) <= 1e-9 * np.linalg.norm(xr)
    # every rank saw the same scalars, and the two exchange forms agree with each other
    for exch in (0, 1):
        assert len({tuple(h[0] for h in out[exch][2]) for _, out in outs}) == 1


# ----------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at bench size: C3D10 twist plate 48 x 6 x 72 cells (124 416 elements, the same 182 845 nodes /
# 548 535 DOF as the 1 M C3D4 plate), rows of 27..65 blocks: row-centric assembly + 4-wavefront SpMV vs the C oracle.
@pytest.fixture(scope="module")
def quad():
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_quadratic_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from femcy_amd.user_defined import user_dirichletBC_values
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    from oracle.femcy_oracle import Material
    from helpers import node_adjacency
    m = meshgen.twist_plate(48, 6, 72, quadratic=True)                     # the mesh bench.py --workload c3d10 runs
    assert m["elements"].shape == (124416, 10) and m["nodes"].shape == (182845, 3)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern()
    u = np.zeros(ctx.n)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
        if bc["user"]:
            user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
    cons = np.unique(np.concatenate(cons))
    ed = elem_def("C3D10")
    ptr, idx = node_adjacency(m["elements"], m["nodes"].shape[0])
    assert idx.size == info.nnzb and info.max_row_blocks == int(np.diff(ptr).max())
    co = COracle(m["nodes"], m["elements"], ed.dN_table(), ed.gauss_weights, Material("lin3d", m["elastic"]).C, ptr, idx)
    yield dict(be=be, ctx=ctx, m=m, u=u, cons=cons, co=co, info=info)
    ctx.close()


def test_c3d10_bench_size_against_c_oracle(quad):
    be, ctx, u, cons, co, m = quad["be"], quad["ctx"], quad["u"], quad["cons"], quad["co"], quad["m"]
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)                                   # AUTO: the two-rows-per-wave kernel (ROWS4) for C3D10
    co.get_dsdx_and_vol(u)
    co.assemble()
    assert rel(ctx.gauss_field(be.GP_VOL).to_numpy(), co.vol) < 1e-12
    assert rel(ctx.gauss_field(be.GP_DSDX).to_numpy(), co.dsdx) < 1e-12
    x = np.random.default_rng(0).standard_normal(ctx.n)
    yo = co.compute_Ad(x)
    ctx.upload(be.VEC_TMP0, x)
    ys = {}
    for wps in (0, 1, 2, 4):                                     # every SpMV variant on the long rows (0 = auto = 4)
        ctx.set_option(be.OPT_SPMV_VARIANT, wps)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        ys[wps] = ctx.download(be.VEC_TMP1)
        assert rel(ys[wps], yo) < 1e-12
    ctx.set_option(be.OPT_SPMV_VARIANT, 0)
    scale = np.abs(yo).max()
    for mode in (be.ASM_GATHER, be.ASM_GATHER_SYM, be.ASM_GATHER_SYM_ROWSUM, be.ASM_ATOMIC, be.ASM_ROWS, be.ASM_ROWS2, be.ASM_ROWS3, be.ASM_ROWS4, be.ASM_AUTO):
        ctx.set_option(be.OPT_ASSEMBLY, mode)
        ctx.assemble_K(be.VEC_DOF)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        assert np.abs(ctx.download(be.VEC_TMP1) - yo).max() < 1e-12 * scale, mode
    ctx.assemble_K(be.VEC_DOF)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    assert np.array_equal(ctx.download(be.VEC_TMP1), ys[0])      # the default assembly is bit-reproducible
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f = co.internal_force(u, 0, *m["elastic"])
    assert rel(ctx.download(be.VEC_FORCE), f) < 1e-11
    assert rel(ctx.gauss_field(be.GP_SIGMA).to_numpy(), co.sigma) < 1e-11
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.assemble_K(be.VEC_DOF)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    co.get_dsdx_and_vol(u)
    co.assemble()
    co.zero_rows_cols_unit_diag(cons)
    f[cons] = 0.0
    assert rel(ctx.download(be.VEC_RESIDUAL), f) < 1e-9
    # both sides iterate on the SAME right-hand side (the device's): the recurrence is what is compared
    b = ctx.download(be.VEC_RESIDUAL)
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
    xo, ito, r0o, rmaxo = co.cg(b, eps=0.0, maxit=30)
    assert it == ito == 30 and abs(r0 - r0o) < 1e-11 * r0o and abs(rmax - rmaxo) < 1e-6 * rmaxo
    assert np.linalg.norm(ctx.download(be.VEC_X) - xo) / np.linalg.norm(xo) < 1e-8
    # properties
    rng = np.random.default_rng(3)

    def Kx(v):
        ctx.upload(be.VEC_TMP0, v)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        return ctx.download(be.VEC_TMP1)

    ctx.assemble_K(be.VEC_DOF)
    z = rng.standard_normal(ctx.n)
    Kx1, Kz1 = Kx(x), Kx(z)
    assert abs(z @ Kx1 - x @ Kz1) < 1e-10 * abs(z @ Kx1)
    for i in range(3):
        t = np.zeros(ctx.n)
        t[i::3] = 1.0
        assert np.abs(Kx(t)).max() < 1e-9 * np.abs(Kx1).max()
    ctx.upload(be.VEC_RESIDUAL, f)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    it1, _, rm1 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
    xs = ctx.download(be.VEC_X)
    assert 0 < it1 < ctx.n and rm1 < 1e-3 * r0
    it2, _, rm2 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
    assert (it2, rm2) == (it1, rm1) and np.array_equal(ctx.download(be.VEC_X), xs)


# ---------------------------------------------------------------------------- BASELINE configs[1] at bench size (round 6)
@pytest.fixture(scope="module")
def beam():
    """the 1280 x 128 plane-strain CPE8 beam of `bench.py --workload cpe8` (163 840 elements, 988 674 DOF) at state S1"""
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    from femcy_amd.material_zoo import LinearIsotropicPlaneStrain
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    from oracle.femcy_oracle import Material
    m = meshgen.beam_quad8(1280, 128, plane="CPE8")
    assert m["elements"].shape == (163840, 8) and m["nodes"].shape == (494337, 2)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_quadrilateral())
    ctx.set_material(LinearIsotropicPlaneStrain(*m["elastic"]))
    info = ctx.build_pattern()
    assert (info.n, info.nnzb, info.max_row_blocks) == (988674, 7711745, 21)
    ti = m["time_incs"]
    u = np.zeros(ctx.n)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        dofs = np.asarray(bc["node_set"]) * 2 + bc["dof"]
        cons.append(dofs)
        u[dofs] = bc["val"] * ti["ini_inc"] / ti["max_time"]
    cons = np.unique(np.concatenate(cons))
    # a smooth interior displacement on top of the prescribed values: the assembly is on a DEFORMED configuration
    x = m["nodes"] / 40.0
    free = np.ones(ctx.n, dtype=bool)
    free[cons] = False
    u += free * 0.05 * np.stack([np.sin(3.1 * x[:, 0]) * np.cos(17.0 * x[:, 1]), np.cos(2.3 * x[:, 0] + 0.3) * x[:, 1] * 10.0], axis=1).ravel()
    ed = elem_def("CPE8")
    co = COracle(m["nodes"], m["elements"], ed.dN_table(), ed.gauss_weights, Material("pstrain", m["elastic"]).C)
    yield dict(be=be, ctx=ctx, m=m, u=u, cons=cons, co=co, info=info)
    ctx.close()


def test_cpe8_bench_size_against_c_oracle(beam):
    """geometry, every assembly variant that exists for 2-D elements (the round-6 pair-list kernel under every value of
    its knobs: the SAME bits), the product, the Newton Dirichlet treatment and 30 PCG iterations against the as-written C
    restatement at the size of `hbm_bound[2]`; symmetry, rigid-body null space, bit-reproducible re-assembly"""
    be, ctx, u, cons, co = beam["be"], beam["ctx"], beam["u"], beam["cons"], beam["co"]
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)                                   # AUTO: FEMCY_ASM_PAIRS
    co.get_dsdx_and_vol(u)
    co.assemble()
    # the last element column carries the first increment's tip displacement (u_y = 5 across a 0.03-wide element: a
    # shear of 160): det J = J00 J11 - J01 J10 there multiplies the rounding noise of J01 (1e-14: coordinates up to 40,
    # sums of eight terms of size 60) by 2.5 -- 3e-10 of det J between ANY two summation orders (measured between the C
    # restatement and the device; 1e-12 at u = 0).  Away from that column: 1e-11 -- the Jacobian of a 0.03-wide element
    # is a difference of coordinates up to 40, a cancellation of 2 560 (measured 1.8e-12; the C3D4 / C3D10 plates, with
    # elements of size 1.4 at coordinates up to 80, hold 1e-12).
    tip = np.zeros(ctx.n // 2, dtype=bool)
    tip[cons // 2] = True
    away = ~tip[beam["m"]["elements"]].any(axis=1)
    assert away.sum() >= 163840 - 2 * 128
    vol, dsdx = ctx.gauss_field(be.GP_VOL).to_numpy(), ctx.gauss_field(be.GP_DSDX).to_numpy()
    assert rel(vol[away], co.vol[away]) < 1e-11 and rel(vol, co.vol) < 1e-9
    assert rel(dsdx[away], co.dsdx[away]) < 1e-11 and rel(dsdx, co.dsdx) < 1e-9
    x = np.random.default_rng(0).standard_normal(ctx.n)
    yo = co.compute_Ad(x)
    scale = np.abs(yo).max()
    # rows of the nodes of that last column: K there is 1e5 times the rest (gradients of the sheared elements) and
    # carries their 3e-10; every other row is held to 1e-11 of the largest entry of K x AMONG those rows
    near = np.zeros(ctx.n // 2, dtype=bool)
    near[beam["m"]["elements"][~away].ravel()] = True
    far = np.repeat(~near, 2)

    def same_product(y, tag):
        assert np.abs(y - yo)[far].max() < 1e-11 * np.abs(yo[far]).max(), tag
        assert np.abs(y - yo).max() < 1e-9 * scale, tag

    ctx.upload(be.VEC_TMP0, x)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    y_auto = ctx.download(be.VEC_TMP1)
    same_product(y_auto, "auto")
    K_auto = ctx.get_K_bsr()
    for mode in (be.ASM_GATHER, be.ASM_GATHER_SYM, be.ASM_GATHER_SYM_ROWSUM, be.ASM_ATOMIC, be.ASM_ROWS, be.ASM_PAIRS):
        ctx.set_option(be.OPT_ASSEMBLY, mode)
        ctx.assemble_K(be.VEC_DOF)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        same_product(ctx.download(be.VEC_TMP1), mode)
    # the knobs of the pair-list kernel move work between waves, never the order in which a block sums its elements
    for knobs in (0, 1, 2, 32, 35, 8 + 35, 16 + 33, 64 * 4 + 35, 64 * 15 + 33, 163):
        ctx.set_option(be.TUNE_PAIRS, knobs)
        ctx.assemble_K(be.VEC_DOF)
        Kk = ctx.get_K_bsr()
        assert np.array_equal(Kk.indices, K_auto.indices) and np.array_equal(Kk.data, K_auto.data), knobs
    ctx.set_option(be.TUNE_PAIRS, -1)
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_AUTO)
    ctx.assemble_K(be.VEC_DOF)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    assert np.array_equal(ctx.download(be.VEC_TMP1), y_auto)
    # symmetry and the rigid translations (pre-Dirichlet matrix)
    z = np.random.default_rng(3).standard_normal(ctx.n)
    ctx.upload(be.VEC_TMP0, z)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    Kz = ctx.download(be.VEC_TMP1)
    assert abs(x @ Kz - z @ y_auto) < 1e-10 * abs(x @ Kz)
    for i in range(2):
        t = np.zeros(ctx.n)
        t[i::2] = 1.0
        ctx.upload(be.VEC_TMP0, t)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        assert np.abs(ctx.download(be.VEC_TMP1)).max() < 1e-9 * scale
    # Newton residual, Dirichlet treatment, 30 iterations of the recurrence
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f = co.internal_force(u, 1, *beam["m"]["elastic"])                # kind 1 = plane strain (femcy_oracle.c: orc_cauchy_large)
    fd = ctx.download(be.VEC_FORCE)
    assert np.abs(fd - f)[far].max() < 1e-10 * np.abs(f[far]).max() and rel(fd, f) < 1e-9
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.assemble_K(be.VEC_DOF)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    co.get_dsdx_and_vol(u)
    co.assemble()
    co.zero_rows_cols_unit_diag(cons)
    f[cons] = 0.0
    assert rel(ctx.download(be.VEC_RESIDUAL), f) < 1e-9
    # both sides iterate on the SAME right-hand side (the device's): the recurrence is what is compared
    b = ctx.download(be.VEC_RESIDUAL)
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
    xo, ito, r0o, rmaxo = co.cg(b, eps=0.0, maxit=30)
    assert it == ito == 30 and abs(r0 - r0o) < 1e-11 * r0o and abs(rmax - rmaxo) < 1e-6 * rmaxo
    assert np.linalg.norm(ctx.download(be.VEC_X) - xo) / np.linalg.norm(xo) < 1e-8


# ------------------------------------------------------------------- C3D10 out of cache (SURVEY 8d "then raise k", round 6)
def test_c3d10_k12_out_of_cache_against_c_oracle():
    """995 328 C3D10 elements, 4 183 275 DOF, 2.99 GB of stored matrix -- twelve times the Infinity Cache, eight times
    every earlier C3D10 record: geometry, the default assembly (k_assemble_rows4) and its launch-order / write-out knobs
    (the same bits), the 4-wave product, the Dirichlet treatment and 30 iterations of the three-launch PCG against the
    as-written C restatement (13 GB of ELL arrays on the host), plus symmetry and the rigid translations."""
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_quadratic_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from femcy_amd.user_defined import user_dirichletBC_values
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    from oracle.femcy_oracle import Material
    from helpers import node_adjacency
    m = meshgen.twist_plate_k(12, quadratic=True)
    assert m["elements"].shape == (995328, 10) and m["nodes"].shape == (1394425, 3)
    ctx = be.Context(0)
    try:
        ctx.set_mesh(m["nodes"], m["elements"])
        ctx.set_element(Element_quadratic_tetrahedral())
        ctx.set_material(LinearIsotropic(*m["elastic"]))
        info = ctx.build_pattern()
        assert (info.n, info.nnzb, info.max_row_blocks) == (4183275, 38924641, 65)
        u = np.zeros(ctx.n)
        cons = []
        for bc in m["dirichlet_bc_info"]:
            cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
            if bc["user"]:
                user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
        cons = np.unique(np.concatenate(cons))
        ed = elem_def("C3D10")
        ptr, idx = node_adjacency(m["elements"], m["nodes"].shape[0])
        assert idx.size == info.nnzb
        co = COracle(m["nodes"], m["elements"], ed.dN_table(), ed.gauss_weights, Material("lin3d", m["elastic"]).C, ptr, idx)
        ctx.upload(be.VEC_DOF, u)
        ctx.assemble_K(be.VEC_DOF)
        co.get_dsdx_and_vol(u)
        co.assemble()
        assert rel(ctx.gauss_field(be.GP_VOL).to_numpy(), co.vol) < 1e-12
        assert rel(ctx.gauss_field(be.GP_DSDX).to_numpy(), co.dsdx) < 1e-12
        x = np.random.default_rng(0).standard_normal(ctx.n)
        yo = co.compute_Ad(x)
        scale = np.abs(yo).max()
        ctx.upload(be.VEC_TMP0, x)
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        y0 = ctx.download(be.VEC_TMP1)
        assert np.abs(y0 - yo).max() < 1e-12 * scale
        for order in (1, 2, 3, 0):                               # launch orders of rows4: work moves between workgroups only
            ctx.set_option(be.TUNE_ROWS4_ORDER, order)
            ctx.assemble_K(be.VEC_DOF)
            ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
            assert np.array_equal(ctx.download(be.VEC_TMP1), y0), order
        ctx.set_option(be.TUNE_ROWS4_ORDER, -1)
        z = np.random.default_rng(3).standard_normal(ctx.n)
        ctx.upload(be.VEC_DU, z)
        ctx.spmv(be.VEC_DU, be.VEC_TMP1)
        Kz = ctx.download(be.VEC_TMP1)
        assert abs(x @ Kz - z @ y0) < 1e-10 * abs(x @ Kz)
        for i in range(3):
            t = np.zeros(ctx.n)
            t[i::3] = 1.0
            ctx.upload(be.VEC_DU, t)
            ctx.spmv(be.VEC_DU, be.VEC_TMP1)
            assert np.abs(ctx.download(be.VEC_TMP1)).max() < 1e-9 * scale
        ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
        f = co.internal_force(u, 0, *m["elastic"])
        assert rel(ctx.download(be.VEC_FORCE), f) < 1e-11
        ctx.vector(be.VEC_RHS).fill(0.0)
        ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
        ctx.assemble_K(be.VEC_DOF)
        ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
        co.get_dsdx_and_vol(u)
        co.assemble()
        co.zero_rows_cols_unit_diag(cons)
        f[cons] = 0.0
        assert rel(ctx.download(be.VEC_RESIDUAL), f) < 1e-11
        it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        xo, ito, r0o, rmaxo = co.cg(f, eps=0.0, maxit=30)
        assert it == ito == 30 and abs(r0 - r0o) < 1e-11 * r0o and abs(rmax - rmaxo) < 1e-6 * rmaxo
        assert np.linalg.norm(ctx.download(be.VEC_X) - xo) / np.linalg.norm(xo) < 1e-8
        tm = ctx.timing()
        assert tm["solves_three"] >= 1 and tm["solves_persist"] == 0     # 21 790 slices: beyond the persistent kernel's layout
    finally:
        ctx.close()
