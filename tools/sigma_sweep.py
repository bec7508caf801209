"""C3D4 plates: SpMV / PCG time against the SELL sorting window.  usage: python tools/sigma_sweep.py k"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
m = meshgen.twist_plate_k(k)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
for sigma in (4096, 32768, 1 << 20):
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    ctx.set_option(be.OPT_SELL_SIGMA, sigma)
    info = ctx.build_pattern()
    spmv_bytes = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
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
    print(f"k {k} sigma {sigma:8d} stored {info.stored_blocks}: spmv {ts*1e6:6.1f} us ({spmv_bytes/ts/1e9:5.0f} GB/s), pcg {tp*1e6:6.1f} us/it", flush=True)
    ctx.close()
