"""SpMV / PCG time against the cap on workgroups per XCD (knob 101) on the bench meshes.
usage: python tools/spmv_cap_probe.py c3d10|c3d4|c3d4_8m"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

wl = sys.argv[1] if len(sys.argv) > 1 else "c3d10"
quad = wl == "c3d10"
m = meshgen.twist_plate(48, 6, 72, quadratic=True) if quad else meshgen.twist_plate_k(24 if wl == "c3d4_8m" else 12)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"])
ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
ctx.set_material(LinearIsotropic(*m["elastic"]))
info = ctx.build_pattern()
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
spmv_bytes = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n
ctx.assemble_K(-1)
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
print(f"{wl}: nslices {info.nslices}")
for cap in (64, 128, 192, 256, 384, 512):
    ctx.set_option(101, cap)
    for _ in range(20):
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(200):
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    ctx.sync()
    ts = (time.perf_counter() - t) / 200
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
    t = time.perf_counter()
    it, _, _ = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=500)
    tp = (time.perf_counter() - t) / it
    print(f"  cap {cap:4d}: spmv {ts*1e6:6.1f} us ({spmv_bytes/ts/1e9:5.0f} GB/s), pcg {tp*1e6:6.1f} us/it", flush=True)
ctx.close()
