"""-m gpu: the persistent one-launch PCG (kernels_pcg_persist.hip; vectors and part of the matrix in registers, part in
LDS, three grid barriers per iteration) against the three-kernel loop and the numpy restatement of the reference
recurrence (conjugateGradientSolver.py:103-127), on meshes large enough to fill the chip (>= 256 slices of 64 nodes)."""
import numpy as np
import pytest

import oracle.femcy_oracle as orc

pytestmark = pytest.mark.gpu


def _system(gpu_ctx_factory, mesh, element):
    from femcy_amd import backend as be
    from femcy_amd.material_zoo import LinearIsotropic, LinearIsotropicPlaneStrain
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh["nodes"], mesh["elements"])
    ctx.set_element(element)
    dm = mesh["nodes"].shape[1]
    ctx.set_material(LinearIsotropic(*mesh["elastic"]) if dm == 3 else LinearIsotropicPlaneStrain(*mesh["elastic"]))
    info = ctx.build_pattern()
    ctx.assemble_K(-1)
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * dm + b["dof"] for b in mesh["dirichlet_bc_info"]]))
    b = np.sin(np.arange(ctx.n) * 0.11) * 1e3
    ctx.upload(be.VEC_RESIDUAL, b)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    return be, ctx, info, b


def _paths(ctx):
    t = ctx.timing()
    return t["solves_three"], t["solves_small"], t["solves_persist"]


@pytest.fixture(scope="module")
def plate(gpu_ctx_factory):
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    m = meshgen.twist_plate(24, 6, 96)                    # 82 944 C3D4, 16 975 nodes = 266 slices, 50 925 DOF
    be, ctx, info, b = _system(gpu_ctx_factory, m, Element_linear_tetrahedral())
    assert 256 <= info.nslices < 384      # fills the chip once: below the default threshold (1.5 x), so PERSIST = 2
    K = ctx.get_K_bsr().tocsr()
    return dict(be=be, ctx=ctx, K=K, b=b, bb=ctx.download(be.VEC_RESIDUAL))


def _solve(ctx, be, eps, maxit):
    r = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=eps, maxit=maxit)
    return r, ctx.download(be.VEC_X)


def test_persistent_pcg_is_the_path_and_equals_the_three_kernel_loop(plate):
    be, ctx, K, bb = plate["be"], plate["ctx"], plate["K"], plate["bb"]
    res = {}
    for persist in (0, 1):
        ctx.set_option(be.OPT_PCG_PERSIST, 2 * persist)
        before = _paths(ctx)
        res[persist] = [_solve(ctx, be, eps, maxit) for eps, maxit in ((0.0, 1), (0.0, 7), (0.0, 40), (1e-10, 10 ** 6))]
        after = _paths(ctx)
        # the path that ran: four solves, all three-kernel or all persistent, never the small-system kernel
        assert (after[0] - before[0], after[1] - before[1], after[2] - before[2]) == ((4, 0, 0), (0, 0, 4))[persist]
    ctx.set_option(be.OPT_PCG_PERSIST, 1)                  # default rule at this size: three launches
    before = _paths(ctx)
    _solve(ctx, be, 0.0, 3)
    assert _paths(ctx)[0] - before[0] == 1
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    for (((it0, r00, rm0), x0), ((it1, r01, rm1), x1)), tol in zip(zip(res[0][:3], res[1][:3]), (1e-13, 1e-12, 1e-10)):
        assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
        assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
    # against the numpy restatement of the reference recurrence
    for ((it, r0, rm), x), maxit, tol in zip(res[1][:3], (1, 7, 40), (1e-13, 1e-12, 1e-10)):
        xo, ito, r0o, rmo = orc.pcg_reference(K, bb, eps=0.0, maxit=maxit)
        assert it == ito == maxit and r0 == r0o and abs(rm - rmo) <= tol * rmo
        assert np.linalg.norm(x - xo) <= tol * np.linalg.norm(xo)
    (it0, r00, rm0), x0 = res[0][3]
    (it1, r01, rm1), x1 = res[1][3]
    assert abs(it0 - it1) <= max(2, it0 // 50) and rm1 < 1e-10 * r01
    assert np.abs(K @ x1 - bb).max() < 2e-10 * r01 + 1e-9 * r01
    assert np.linalg.norm(x1 - x0) <= 1e-7 * np.linalg.norm(x0)
    # run-to-run bit reproducible (fixed summation order, fixed slice-to-wave assignment)
    (it2, _, rm2), x2 = _solve(ctx, be, 1e-10, 10 ** 6)
    assert it2 == it1 and rm2 == rm1 and np.array_equal(x2, x1)


@pytest.mark.parametrize("knobs", [dict(rj=0, lds=0), dict(rj=0, lds=-1), dict(rj=4, lds=0), dict(rj=5, lds=3),
                                   dict(rj=4, lds=-1, dbg=16)])
def test_persistent_pcg_residency_variants_agree(plate, knobs):
    """where a block row lives (registers, LDS, streamed, prefetched during the barriers) changes the order of a row's
    partial products, not the recurrence: iterates agree to rounding, the converged solution solves the system"""
    be, ctx, K, bb = plate["be"], plate["ctx"], plate["K"], plate["bb"]
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    (itr, _, rmr), xr = _solve(ctx, be, 0.0, 25)
    ctx.set_option(105, knobs["rj"])
    ctx.set_option(104, knobs["lds"])
    ctx.set_option(106, knobs.get("dbg", 0))
    try:
        before = _paths(ctx)
        (it, r0, rm), x = _solve(ctx, be, 0.0, 25)
        assert it == itr == 25 and abs(rm - rmr) <= 1e-11 * rmr and np.linalg.norm(x - xr) <= 1e-11 * np.linalg.norm(xr)
        (it, r0, rm), x = _solve(ctx, be, 1e-8, 10 ** 6)
        assert rm < 1e-8 * r0 and np.abs(K @ x - bb).max() < 2.1e-8 * r0
        assert _paths(ctx)[2] - before[2] == 2
    finally:
        ctx.set_option(105, 4)
        ctx.set_option(104, -1)
        ctx.set_option(106, 0)


def test_persistent_pcg_edge_cases(plate):
    be, ctx, b = plate["be"], plate["ctx"], plate["b"]
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    before = _paths(ctx)
    try:
        ctx.vector(be.VEC_RESIDUAL).fill(0.0)                           # b = 0: zero iterations, x = 0
        it, r0, rm = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        assert it == 0 and r0 == 0.0 and not ctx.download(be.VEC_X).any()
        bad = plate["bb"].copy()
        bad[5] = np.nan
        ctx.upload(be.VEC_RESIDUAL, bad)
        with pytest.raises(be.FemcyError) as ei:
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        assert ei.value.status == be.FEMCY_ENUMERIC
        bad[5] = np.inf
        ctx.upload(be.VEC_RESIDUAL, bad)
        with pytest.raises(be.FemcyError) as ei:
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        assert ei.value.status == be.FEMCY_ENUMERIC
        assert _paths(ctx)[2] - before[2] == 3
        # an eps that the initial residual already meets after one iteration at the latest; r0 is max|b|
        ctx.upload(be.VEC_RESIDUAL, plate["bb"])
        it, r0, rm = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=10.0)
        assert it == 1 and r0 == np.abs(plate["bb"]).max() and rm < 10.0 * r0
    finally:
        ctx.upload(be.VEC_RESIDUAL, plate["bb"])


def test_barrier_timeout_falls_back_to_the_three_kernel_loop(plate):
    """more workgroups than can be co-resident (two per CU at 512 registers per lane).  With the co-residency check
    (round 3) the grid is refused before the launch; with the check switched off the first grid barrier can never
    complete: the bounded spin poisons the counter, every workgroup leaves, and the solve is redone by the three-kernel
    loop -- same answer, and the context does not try the persistent kernel again.  (Round 2's version of this test
    never launched the over-sized grid: 266 slices < 512 workgroups failed the eligibility test first; the
    `barrier_timeouts` counter now proves the time-out happened.)"""
    be, ctx = plate["be"], plate["ctx"]
    ctx.set_option(be.OPT_PCG_PERSIST, 0)
    (it0, r00, rm0), x0 = _solve(ctx, be, 0.0, 12)
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    ctx.set_option(107, 512)
    # with the co-residency check (round 3) the over-sized grid is refused BEFORE the launch: three launches, no time-out
    before, t0 = _paths(ctx), ctx.timing()["barrier_timeouts"]
    (itc, r0c, rmc), xc = _solve(ctx, be, 0.0, 12)
    after = _paths(ctx)
    assert (after[0] - before[0], after[2] - before[2]) == (1, 0) and ctx.timing()["barrier_timeouts"] == t0
    assert (itc, r0c, rmc) == (it0, r00, rm0) and np.array_equal(xc, x0)
    ctx.set_option(be.TUNE_SKIP_OCCUPANCY_CHECK, 1)                     # ... and without the check it times out
    try:
        before = _paths(ctx)
        (it1, r01, rm1), x1 = _solve(ctx, be, 0.0, 12)
        after = _paths(ctx)
        assert (after[0] - before[0], after[2] - before[2]) == (1, 0)
        assert (it1, r01, rm1) == (it0, r00, rm0) and np.array_equal(x1, x0)
        (it2, _, _), x2 = _solve(ctx, be, 0.0, 12)                      # no second attempt (no second time-out)
        assert _paths(ctx)[0] - after[0] == 1 and np.array_equal(x2, x0)
        assert ctx.timing()["barrier_timeouts"] == t0 + 1
    finally:
        ctx.set_option(be.TUNE_SKIP_OCCUPANCY_CHECK, 0)
        ctx.set_option(107, 0)                                          # also clears the "failed once" mark
    before = _paths(ctx)
    _solve(ctx, be, 0.0, 12)
    assert _paths(ctx)[2] - before[2] == 1


def test_persistent_pcg_two_dimensional_blocks(gpu_ctx_factory):
    """dm = 2 instantiation (2 x 2 blocks) on a CPE8 beam of 65 k nodes"""
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    m = meshgen.beam_quad8(nx=360, ny=60)                               # 21 600 CPE8, 65 761 nodes
    be, ctx, info, b = _system(gpu_ctx_factory, m, Element_quadratic_quadrilateral())
    assert info.nslices >= 256
    out = {}
    for persist in (0, 1):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        before = _paths(ctx)
        out[persist] = [_solve(ctx, be, 0.0, k) for k in (1, 9, 30)]
        assert _paths(ctx)[2 if persist else 0] - before[2 if persist else 0] == 3
    for (((it0, r00, rm0), x0), ((it1, r01, rm1), x1)), tol in zip(zip(out[0], out[1]), (1e-13, 1e-11, 1e-9)):
        assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
        assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
    ctx.close()


@pytest.mark.parametrize("cells,spw", [((1000, 100), 6), ((1280, 128), 8)])
def test_persistent_pcg_2d_six_and_eight_slices_per_wave(gpu_ctx_factory, cells, spw):
    """round 5: with 2 x 2 blocks a wave has the registers for up to EIGHT slices (vectors 20 registers per slice, a
    block row 9), so 2-D systems of up to 8 192 slices = 1.05 M DOF take the one-launch path: the CPE8 beam of
    `bench.py --workload cpe8` (BASELINE configs[1] at the size of the 3-D headline system, 7 725 slices, SPW = 8) and a
    4 097 ... 6 144-slice one (SPW = 6).  Same recurrence: iterates equal the three-launch loop's to rounding."""
    import time
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    m = meshgen.beam_quad8(*cells)
    be, ctx, info, b = _system(gpu_ctx_factory, m, Element_quadratic_quadrilateral())
    lo, hi = {6: (4096, 6144), 8: (6144, 8192)}[spw]
    assert lo < info.nslices <= hi
    out, us = {}, {}
    for persist in (0, 1):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        before = _paths(ctx)
        out[persist] = [_solve(ctx, be, 0.0, k) for k in (1, 9, 30)]
        assert _paths(ctx)[2 if persist else 0] - before[2 if persist else 0] == 3
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
        t = time.perf_counter()
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=300)
        us[persist] = (time.perf_counter() - t) / 300 * 1e6
    for (((it0, r00, rm0), x0), ((it1, r01, rm1), x1)), tol in zip(zip(out[0], out[1]), (1e-13, 1e-11, 1e-9)):
        assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
        assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
    again = _solve(ctx, be, 0.0, 30)
    assert np.array_equal(again[1], out[1][2][1])                      # a solve is bit-reproducible
    print(f"[CPE8 {cells[0]}x{cells[1]}: {info.nslices} slices, {spw} per wave] three launches {us[0]:.1f} us / iteration, persistent {us[1]:.1f}")
    assert us[1] < us[0]                                               # the default path is the faster one
    ctx.close()


def test_persistent_pcg_four_slices_per_wave(gpu_ctx_factory):
    """above 3 x 128 slices per XCD range a wave owns up to FOUR slices (SPW = 4 instantiation, 3 register rows per
    slice): a 1.09 M-element C3D4 plate (200 889 nodes = 3 139 slices, 217 MB of matrix: still Infinity-Cache size).
    And the 124 k C3D10 plate (380 MB; 287 MB of it streamed from HBM every iteration) IS taken by default since round 5:
    the 240 MiB rule of rounds 2-4 rested on a round-2 measurement and was stale (profiles/r05_persist_hbm_c3d10.txt)."""
    import time
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
    m = meshgen.twist_plate(100, 12, 152)
    be, ctx, info, b = _system(gpu_ctx_factory, m, Element_linear_tetrahedral())
    assert 3072 < info.nslices <= 4096
    out, us = {}, {}
    for persist in (0, 1):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        before = _paths(ctx)
        out[persist] = [_solve(ctx, be, 0.0, k) for k in (1, 9, 30)]
        assert _paths(ctx)[2 if persist else 0] - before[2 if persist else 0] == 3
        t = time.perf_counter()
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=300)
        us[persist] = (time.perf_counter() - t) / 300 * 1e6
    for (((it0, r00, rm0), x0), ((it1, r01, rm1), x1)), tol in zip(zip(out[0], out[1]), (1e-13, 1e-12, 1e-10)):
        assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
        assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
    print(f"[1.09 M C3D4, 4 slices per wave] three launches {us[0]:.1f} us / iteration, persistent {us[1]:.1f}")
    assert us[1] < us[0]                                   # the form that is chosen by default is the faster one
    ctx.close()
    mq = meshgen.twist_plate(48, 6, 72, quadratic=True)
    be, ctx, info, b = _system(gpu_ctx_factory, mq, Element_quadratic_tetrahedral())
    before = _paths(ctx)
    ctx.set_option(be.TUNE_PERSIST_MAX_MB, 240)            # the rule of rounds 2-4: streamed part within the Infinity Cache
    (it0, r00, rm0), x0 = _solve(ctx, be, 0.0, 20)
    assert _paths(ctx)[0] - before[0] == 1                 # ... three launches for this matrix
    ctx.set_option(be.TUNE_PERSIST_MAX_MB, 0)              # default since round 5: no byte limit (61 against 78 us / iteration)
    (it1, r01, rm1), x1 = _solve(ctx, be, 0.0, 20)
    assert _paths(ctx)[2] - before[2] == 1
    assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= 1e-11 * rm0
    assert np.linalg.norm(x1 - x0) <= 1e-11 * np.linalg.norm(x0)
    us = {}
    for mb in (240, 0):
        ctx.set_option(be.TUNE_PERSIST_MAX_MB, mb)
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
        best = 1e30
        for _ in range(3):       # best of three: numpy's BLAS pool, still spinning after the norms above, stalls the host
            t = time.perf_counter()                      # for 35-75 ms now and then (one such stall = +150 us / iteration)
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=300)
            best = min(best, (time.perf_counter() - t) / 300 * 1e6)
        us[mb] = best
    print(f"[124 k C3D10, matrix streamed from HBM] three launches {us[240]:.1f} us / iteration, persistent {us[0]:.1f}")
    assert us[0] < us[240]                                 # the form that is chosen by default is the faster one
    ctx.close()


# ------------------------------------------------------------------------------------------------ round 3
@pytest.mark.parametrize("var", [0, 1, 2, 3, 4, 5, 6, 7])
def test_persistent_pcg_variants_agree(plate, var):
    """FEMCY_TUNE_PERSIST_VARIANT: sweep direction, exchange form and the cache policy of the matrix stream and the order in which d is published change
    the order of partial sums and nothing else -- iterates agree with the round-2 form to rounding, the converged solution
    solves the system, a solve is bit-reproducible.  The shipped library holds the default variant and 0; the others
    run when FEMCY_HIP_LIB points at a -DFEMCY_PERSIST_ALL_VARIANTS build."""
    be, ctx, K, bb = plate["be"], plate["ctx"], plate["K"], plate["bb"]
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    ctx.set_option(be.TUNE_PERSIST_VARIANT, 0)
    ref = [_solve(ctx, be, 0.0, k) for k in (1, 8, 25)]
    ctx.set_option(be.TUNE_PERSIST_VARIANT, var)
    try:
        before, t0 = _paths(ctx), ctx.timing()["barrier_timeouts"]
        try:
            got = [_solve(ctx, be, 0.0, k) for k in (1, 8, 25)]
        except be.FemcyError as e:
            if "not in this build" in str(e):
                pytest.skip(str(e))
            raise
        for ((it0, r00, rm0), x0), ((it1, r01, rm1), x1), tol in zip(ref, got, (1e-13, 1e-12, 1e-11)):
            assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
            assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
        (it, r0, rm), x = _solve(ctx, be, 1e-9, 10 ** 6)
        assert rm < 1e-9 * r0 and np.abs(K @ x - bb).max() < 2.1e-9 * r0
        (it2, _, rm2), x2 = _solve(ctx, be, 1e-9, 10 ** 6)
        assert it2 == it and rm2 == rm and np.array_equal(x2, x)
        assert _paths(ctx)[2] - before[2] == 5 and ctx.timing()["barrier_timeouts"] == t0
        # edge cases through this variant's exchanges: b = 0, NaN
        ctx.vector(be.VEC_RESIDUAL).fill(0.0)
        it, r0, rm = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        assert it == 0 and r0 == 0.0 and not ctx.download(be.VEC_X).any()
        bad = bb.copy()
        bad[7] = np.nan
        ctx.upload(be.VEC_RESIDUAL, bad)
        with pytest.raises(be.FemcyError) as ei:
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        assert ei.value.status == be.FEMCY_ENUMERIC
    finally:
        ctx.upload(be.VEC_RESIDUAL, bb)
        ctx.set_option(be.TUNE_PERSIST_VARIANT, -1)


@pytest.mark.parametrize("var", [0, 2])
def test_exchange_timeout_of_every_form_falls_back(plate, var):
    """a spin limit of 0 makes the first exchange of the launch give up (some workgroup always polls before the last
    one has arrived): the poison reaches every workgroup through the form's own channel (top counter / granule tags),
    the solve is redone by the three-kernel loop, the context stops trying"""
    be, ctx = plate["be"], plate["ctx"]
    ctx.set_option(be.OPT_PCG_PERSIST, 0)
    (it0, r00, rm0), x0 = _solve(ctx, be, 0.0, 12)
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    ctx.set_option(be.TUNE_PERSIST_VARIANT, var)
    ctx.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 0)
    try:
        t0 = ctx.timing()["barrier_timeouts"]
        before = _paths(ctx)
        try:
            (it1, r01, rm1), x1 = _solve(ctx, be, 0.0, 12)
        except be.FemcyError as e:
            if "not in this build" in str(e):
                pytest.skip(str(e))
            raise
        after = _paths(ctx)
        assert (after[0] - before[0], after[2] - before[2]) == (1, 0)
        assert ctx.timing()["barrier_timeouts"] == t0 + 1
        assert (it1, r01, rm1) == (it0, r00, rm0) and np.array_equal(x1, x0)
    finally:
        ctx.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 1 << 20)             # also clears the "failed once" marks
        ctx.set_option(be.TUNE_PERSIST_VARIANT, -1)
    before = _paths(ctx)
    _solve(ctx, be, 0.0, 12)
    assert _paths(ctx)[2] - before[2] == 1


def test_small_system_pcg_timeout_falls_back(gpu_ctx_factory):
    """k_pcg_small's grid barrier has the poison-on-time-out protocol of the persistent kernel; a
    time-out is not an error any more -- the three-kernel loop redoes the solve and the context remembers"""
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    m = meshgen.twist_plate(8, 2, 12)
    be, ctx, info, b = _system(gpu_ctx_factory, m, Element_linear_tetrahedral())
    assert info.nslices <= 128 and ctx.n <= 12288
    (it0, r00, rm0), x0 = _solve(ctx, be, 0.0, 15)
    assert _paths(ctx)[1] == 1                                          # the one-launch small-system kernel ran
    ctx.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 0)
    (it1, r01, rm1), x1 = _solve(ctx, be, 0.0, 15)
    p = _paths(ctx)
    assert p[0] == 1 and p[1] == 1 and ctx.timing()["barrier_timeouts"] == 1
    assert it1 == it0 and r01 == r00 and abs(rm1 - rm0) <= 1e-10 * rm0
    assert np.linalg.norm(x1 - x0) <= 1e-10 * np.linalg.norm(x0)
    _solve(ctx, be, 0.0, 15)                                            # remembered: no second attempt
    assert _paths(ctx)[0] == 2 and ctx.timing()["barrier_timeouts"] == 1
    ctx.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 1 << 20)                 # clears the mark
    _solve(ctx, be, 0.0, 15)
    assert _paths(ctx)[1] == 2
    ctx.close()


def test_ceiling_probes(plate):
    """femcy_probe_exchange checks its own sums (every workgroup must have seen every other's value in every round);
    femcy_probe_stream returns a rate; femcy_persist_streamed_bytes is below the stored matrix"""
    be, ctx = plate["be"], plate["ctx"]
    for form in (0, 1):
        us = ctx.probe_exchange(500, form)
        assert 0.2 < us < 50.0, (form, us)
        print(f"[probe] exchange form {form}: {us:.2f} us")
    for mode in (0, 1, 2, 3):
        gbs, moved = ctx.probe_stream(64 << 20, 10, mode)
        assert moved > (60 << 20) and 500.0 < gbs < 40000.0, (mode, gbs, moved)
        print(f"[probe] stream 64 MiB mode {mode}: {gbs:.0f} GB/s")
    info = ctx.pattern_info()
    sb = ctx.persist_streamed_bytes()
    assert 0 <= sb < info.stored_blocks * 76


# ------------------------------------------------------------------------------------------------ round 4
def test_fused_vector_update_equals_the_two_kernels_and_falls_back(plate):
    """FEMCY_OPT_PCG_FUSED_UPDATE: the three-launch loop with ONE vector kernel per iteration (r update, in-kernel
    exchange of (r.M.r, max|r|), d and x update) runs the same recurrence as the two kernels: iteration counts equal,
    iterates to rounding (the reduction is grouped by workgroup in both, in a different order), bit-reproducible.  A spin
    limit of 0 makes the exchange give up: the solve is redone by the two kernels -- same answer -- the time-out is
    counted and the context stays on the two kernels."""
    be, ctx, K, bb = plate["be"], plate["ctx"], plate["K"], plate["bb"]
    ctx.set_option(be.OPT_PCG_PERSIST, 0)
    ctx.set_option(107, 0)
    res = {}
    for flag in (0, 1):
        ctx.set_option(be.OPT_PCG_FUSED_UPDATE, flag)
        res[flag] = [_solve(ctx, be, eps, maxit) for eps, maxit in ((0.0, 1), (0.0, 7), (0.0, 40), (1e-10, 10 ** 6))]
    for (((it0, r00, rm0), x0), ((it1, r01, rm1), x1)), tol in zip(zip(res[0][:3], res[1][:3]), (1e-13, 1e-12, 1e-10)):
        assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
        assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
    (it0, _, _), x0 = res[0][3]
    (it1, r01, rm1), x1 = res[1][3]
    assert abs(it0 - it1) <= max(2, it0 // 50) and rm1 < 1e-10 * r01 and np.abs(K @ x1 - bb).max() < 1.2e-9 * r01
    (it2, _, rm2), x2 = _solve(ctx, be, 1e-10, 10 ** 6)
    assert it2 == it1 and rm2 == rm1 and np.array_equal(x2, x1)
    # the time-out path
    t0 = ctx.timing()
    try:
        ctx.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 0)
        (itb, r0b, rmb), xb = _solve(ctx, be, 0.0, 40)
        (itc, _, rmc), xc = _solve(ctx, be, 0.0, 40)               # stays on the two kernels: no second time-out
        t1 = ctx.timing()
        assert t1["barrier_timeouts"] - t0["barrier_timeouts"] == 1 and t1["solves_three"] - t0["solves_three"] == 2
        (it0, _, rm0), x0 = res[0][2]
        assert itb == itc == it0 == 40 and rmb == rmc == rm0 and np.array_equal(xb, x0) and np.array_equal(xc, x0)
    finally:
        ctx.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 1 << 20)
        ctx.set_option(be.OPT_PCG_FUSED_UPDATE, 0)                 # the default (the fused form measured slower)
        ctx.set_option(be.OPT_PCG_PERSIST, 1)


def test_probe_spmv_times_the_product_in_both_orders(plate):
    """femcy_probe_spmv: launch-to-launch time of the product on the PCG's own vectors, node order and storage order;
    does not disturb the next solve"""
    be, ctx = plate["be"], plate["ctx"]
    ctx.set_option(be.OPT_PCG_PERSIST, 0)
    ref = _solve(ctx, be, 0.0, 9)
    for order in (False, True):
        us = ctx.probe_spmv(50, order)
        assert 0.5 < us < 500.0, us
    got = _solve(ctx, be, 0.0, 9)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    with pytest.raises(be.FemcyError):
        ctx.probe_spmv(0, True)
    ctx.set_option(be.OPT_PCG_PERSIST, 1)


@pytest.mark.parametrize("cells,spw", [((56, 7, 84), 5), ((60, 8, 90), 6), ((64, 8, 96), 7)])
def test_persistent_pcg_five_to_seven_slices_per_wave(gpu_ctx_factory, cells, spw):
    """round 6: 3 x 3 blocks with 5, 6 and 7 slices per wave (one block row per slice in registers, none at 7) -- C3D10
    plates of 0.86 / 1.12 / 1.27 M DOF (k = 7, 7.5, 8: matrices of 0.6 ... 0.9 GB streamed from HBM every iteration) keep ONE
    launch per solve.  Iterates against the three-launch loop at fixed counts (this is the test that caught the 6- and
    7-slice shapes hipcc 7.0 first miscompiled: register copies placed before the exec restore at the join of a masked
    load, tools/check_exec_joins.py); the default path must be the faster one."""
    import time
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_quadratic_tetrahedral
    m = meshgen.twist_plate(*cells, quadratic=True)
    be, ctx, info, b = _system(gpu_ctx_factory, m, Element_quadratic_tetrahedral())
    assert 1024 * (spw - 1) < info.nslices <= 1024 * spw
    out, us = {}, {}
    for persist in (0, 1):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        before = _paths(ctx)
        out[persist] = [_solve(ctx, be, 0.0, k) for k in (1, 9, 30)]
        assert _paths(ctx)[2 if persist else 0] - before[2 if persist else 0] == 3      # the path that ran
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=50)
        best = 1e30
        for _ in range(3):
            t = time.perf_counter()
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=200)
            best = min(best, (time.perf_counter() - t) / 200 * 1e6)
        us[persist] = best
    for (((it0, r00, rm0), x0), ((it1, r01, rm1), x1)), tol in zip(zip(out[0], out[1]), (1e-13, 1e-12, 1e-10)):
        assert it0 == it1 and r00 == r01 and abs(rm0 - rm1) <= tol * rm0
        assert np.linalg.norm(x1 - x0) <= tol * np.linalg.norm(x0)
    iter_bytes = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n + 8 * 11 * ctx.n     # product + 11 vector passes
    print(f"[C3D10 {cells}: {ctx.n} DOF, {info.nslices} slices, {spw} per wave] three launches {us[0]:.1f} us / iteration, "
          f"persistent {us[1]:.1f} = {iter_bytes / us[1] / 1e3 / 8000:.2f} of HBM")
    assert us[1] < us[0]
    # a second solve on the same context: the same bits (no stale state between launches)
    ctx.set_option(be.OPT_PCG_PERSIST, 1)
    r1, x1 = _solve(ctx, be, 0.0, 30)
    r2, x2 = _solve(ctx, be, 0.0, 30)
    assert r1 == r2 and np.array_equal(x1, x2)
    ctx.close()


def test_beyond_seven_slices_per_wave_takes_three_launches(gpu_ctx_factory):
    """a C3D10 plate of 1.6 M DOF (8 slices per wave) is beyond the admitted shapes of the persistent kernel: femcy_pcg
    must take the three-launch loop there by itself (and say so in femcy_timing)"""
    from femcy_amd import meshgen
    from femcy_amd.element_zoo import Element_quadratic_tetrahedral
    be, ctx, info, b = _system(gpu_ctx_factory, meshgen.twist_plate(70, 9, 100, quadratic=True), Element_quadratic_tetrahedral())
    assert 7168 < info.nslices
    before = _paths(ctx)
    (it, r0, rm), x = _solve(ctx, be, 0.0, 5)
    assert it == 5 and np.isfinite(x).all() and _paths(ctx)[0] - before[0] == 1 and _paths(ctx)[2] == before[2]
    ctx.close()
