"""us per PCG iteration on the bench meshes: persistent one-launch kernel (with / without the LDS-resident part of the
matrix) vs the three-kernel loop.  usage: python tools/persist_probe.py c3d4|c3d10 [iters=500]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

wl = sys.argv[1] if len(sys.argv) > 1 else "c3d4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
quad = wl == "c3d10"
m = meshgen.twist_plate(48, 6, 72, quadratic=True) if quad else meshgen.twist_plate_k(12)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"])
ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
ctx.set_material(LinearIsotropic(*m["elastic"]))
info = ctx.build_pattern()
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.assemble_K(-1)
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
bytes_it = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n + 88 * ctx.n
print(f"{wl}: n = {ctx.n}, nslices = {info.nslices}")
ref = None
for label, persist, lds in (("three kernels", 0, -1), ("persistent, no LDS rows", 1, 0), ("persistent, LDS rows auto", 1, -1)):
    ctx.set_option(be.OPT_PCG_PERSIST, persist)
    ctx.set_option(104, lds)
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=50)
    t = time.perf_counter()
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=iters)
    dt = time.perf_counter() - t
    x = ctx.download(be.VEC_X)
    if ref is None:
        ref = x
    err = np.linalg.norm(x - ref) / np.linalg.norm(ref)
    print(f"  {label:<28} {it:5d} iterations, {dt/it*1e6:7.2f} us/iteration ({bytes_it/(dt/it)/1e9:6.0f} GB/s algorithmic), "
          f"rmax = {rmax:.6e}, |x - x_3k| / |x_3k| = {err:.2e}", flush=True)
ctx.close()
