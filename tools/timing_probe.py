"""how well do in-process HIP events reproduce the rocprofv3 kernel duration of k_spmv?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
m = meshgen.twist_plate_k(12)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"]); ctx.set_element(Element_linear_tetrahedral()); ctx.set_material(LinearIsotropic(*m["elastic"]))
ctx.build_pattern(); ctx.assemble_K(-1)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.upload(be.VEC_RESIDUAL, np.ones(ctx.n)); ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
for fence in (0, 1):
    ctx.set_option(100, fence)
    for mode in (1, 16):
        ctx.set_option(be.OPT_TIMING, mode); ctx.timing_reset()
        for _ in range(400): ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        t = ctx.timing(); a = t["spmv_ms"] / t["spmv_launches"] * 1e3
        ctx.timing_reset()
        t0 = time.perf_counter(); ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=500); dt = time.perf_counter() - t0
        t = ctx.timing(); b = t["spmv_ms"] / t["spmv_launches"] * 1e3
        print(f"fence={fence} timing={mode}: standalone spmv {a:.2f} us; inside pcg {b:.2f} us ({t['spmv_launches']} samples), pcg {dt/500*1e6:.1f} us/it")
ctx.set_option(be.OPT_TIMING, 0)
t0 = time.perf_counter(); ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=500); dt = time.perf_counter() - t0
print(f"untimed pcg {dt/500*1e6:.1f} us/it")
