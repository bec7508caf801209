"""GPU micro-benchmarks of the hot path on the synthetic twist plate (prints one line per probe).
usage: python tools/microbench.py [k=12] [quadratic=0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    quad = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    t0 = time.time()
    renum = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    m = meshgen.twist_plate_k(k, quadratic=quad, renumber=renum)
    print(f"mesh k={k} quad={quad}: {m['elements'].shape[0]} elements, {m['nodes'].shape[0]} nodes  ({time.time()-t0:.2f}s)")
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    t0 = time.time()
    info = ctx.build_pattern()
    print(f"pattern: nnzb={info.nnzb} stored={info.stored_blocks} maxrow={info.max_row_blocks} ({time.time()-t0:.2f}s)")
    n, ne = ctx.n, ctx.ne
    nnz = info.nnz
    spmv_bytes = 8 * nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * n
    import torch

    def timeit(fn, reps):
        fn(); ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t) / reps

    # state S1: prescribed twist at t = 0.05 on the z=0 face
    from femcy_amd.user_defined import user_api
    u = np.zeros(n)
    tw = m["node_sets"]["fit_right_z"]
    for d in range(3):
        user_api.user_dirichletBC_values(u, tw, 3, d, m["nodes"], 0.05)
    ctx.upload(be.VEC_DOF, u)
    for mode, nm in ((be.ASM_GATHER, "gather"), (be.ASM_GATHER_SYM, "gather-sym"), (be.ASM_GATHER_SYM_ROWSUM, "gather-sym-rowsum"), (be.ASM_ROWS, "rows"),
                     (be.ASM_ROWS2, "rows2"), (be.ASM_ROWS3, "rows3"), (be.ASM_ROWS4, "rows4"), (be.ASM_ATOMIC, "atomic")):
        if mode == be.ASM_ROWS4 and not quad:
            continue                                   # instantiated for C3D10
        ctx.set_option(be.OPT_ASSEMBLY, mode)
        t = timeit(lambda: ctx.assemble_K(be.VEC_DOF), 20)
        print(f"assemble_K[{nm}]: {t*1e3:.3f} ms  -> {ne/t/1e6:.1f} M elem/s")
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_AUTO)
    ctx.set_option(be.OPT_TIMING, 1)
    ctx.timing_reset()
    for _ in range(10):
        ctx.assemble_K(be.VEC_DOF)
    tm = ctx.timing()
    print(f"  geom kernel {tm['geom_ms']/tm['geom_launches']*1e3:.1f} us, assemble kernel {tm['assemble_ms']/tm['assemble_launches']*1e3:.1f} us")
    ctx.set_option(be.OPT_TIMING, 0)
    t = timeit(lambda: ctx.internal_force(be.VEC_DOF, be.VEC_FORCE), 20)
    print(f"internal_force: {t*1e3:.3f} ms")
    # residual + Newton Dirichlet -> CG rhs
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    cons = np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]])
    ctx.assemble_K(be.VEC_DOF)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    x = np.random.default_rng(0).standard_normal(n)
    ctx.upload(be.VEC_TMP0, x)
    for wps in (1, 2, 4, 0):
        ctx.set_option(be.OPT_SPMV_VARIANT, wps)
        t = timeit(lambda: ctx.spmv(be.VEC_TMP0, be.VEC_TMP1), 200)
        print(f"spmv wps={wps}: {t*1e6:.1f} us  -> {spmv_bytes/t/1e9:.0f} GB/s algorithmic ({spmv_bytes/1e6:.1f} MB)")
    for poll in (32, 128):
        ctx.set_option(be.OPT_PCG_POLL, poll)
        t0 = time.perf_counter()
        it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=500)
        dt = time.perf_counter() - t0
        print(f"pcg poll={poll}: {it} iters in {dt*1e3:.2f} ms -> {it/dt:.0f} it/s, {dt/it*1e6:.1f} us/it, "
              f"{(spmv_bytes + 88*n)*it/dt/1e9:.0f} GB/s algorithmic")
    for cap in (128, 256, 512, 1024, 2048, 4096):
        ctx.set_option(be.OPT_EW_GRID, cap)
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=50)
        t0 = time.perf_counter()
        it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=500)
        dt = time.perf_counter() - t0
        print(f"pcg ew_cap={cap}: {dt/it*1e6:.2f} us/it")
    ctx.set_option(be.OPT_EW_GRID, 2048)
    t0 = time.perf_counter()
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
    dt = time.perf_counter() - t0
    print(f"pcg eps=1e-3: {it} iters, r0={r0:.3e} rmax={rmax:.3e}, {dt*1e3:.1f} ms")


if __name__ == "__main__":
    main()
