"""-m gpu: regressions found by stress runs (one test per root cause)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_zero_fills_have_landed_before_the_first_kernel_of_a_context():
    """Round 5's unexplained "NaN after 1 iteration" (reproduced and bisected in round 6, tools/r06_nan_hunt.py,
    profiles/r06_nan_hunt.txt): femcy_build_pattern zero-filled K with hipMemset -- asynchronous on the NULL stream -- while
    the context's stream is non-blocking, i.e. not ordered behind it.  When the null stream was late (here: after
    the sequence of events below), the fill ran after / during the first assembly and wiped rows the kernel had just
    written: empty rows of K -> 1 / 0 in the Jacobi vector -> NaN.  Every fill of the library now lands before the
    call returns (ctx.hpp: dfill_sync).  The same for the zeroed work buffers of femcy_set_mesh."""
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    import test_gpu_multirank as mr
    # the order of events that made the fill late in every one of 10 trials (tools/records/r06_gpu13.sh; either half
    # alone does not): in-process ranks on threads, then a 1.09 M-element context solved on both PCG paths and closed
    mr.test_neighbour_exchange_equals_allreduce("twist_plate_C3D4.inp", 4, 2)
    m4 = meshgen.twist_plate(100, 12, 152)
    c4 = be.Context(0)
    try:
        c4.set_mesh(m4["nodes"], m4["elements"])
        c4.set_element(Element_linear_tetrahedral())
        c4.set_material(LinearIsotropic(*m4["elastic"]))
        c4.build_pattern()
        c4.assemble_K(-1)
        cons4 = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m4["dirichlet_bc_info"]]))
        c4.upload(be.VEC_RESIDUAL, np.sin(np.arange(c4.n) * 0.11) * 1e3)
        c4.dirichlet_newton(cons4, be.VEC_RESIDUAL)
        for persist in (0, 1):
            c4.set_option(be.OPT_PCG_PERSIST, persist)
            assert c4.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=9)[0] == 9
    finally:
        c4.close()
    m = meshgen.twist_plate(48, 6, 72, quadratic=True)
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    ref = None
    for round_ in range(4):
        ctx = be.Context(0)
        try:
            ctx.set_mesh(m["nodes"], m["elements"])
            ctx.set_element(Element_quadratic_tetrahedral())
            ctx.set_material(LinearIsotropic(*m["elastic"]))
            ctx.build_pattern()                                   # zero-fills K ...
            ctx.assemble_K(-1)                                    # ... which the first kernel of the context stream writes
            K = ctx.get_K_bsr().tocsr()
            rowabs = np.asarray(abs(K).sum(axis=1)).ravel()
            assert (rowabs == 0).sum() == 0 and (K.diagonal() == 0).sum() == 0
            ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
            ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
            ctx.set_option(be.TUNE_PERSIST_MAX_MB, 240)           # the three-launch loop: the call that returned the NaN
            r = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20)
            x = ctx.download(be.VEC_X)
            assert r[0] == 20 and np.isfinite(x).all()
            if ref is None:
                ref = (r, x)
            assert r == ref[0] and np.array_equal(x, ref[1])      # the same bits every round
        finally:
            ctx.close()
