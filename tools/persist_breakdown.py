"""where the time of one iteration of the persistent PCG goes: the same 500-iteration solve with parts of the kernel
switched off through the test knobs 104 (LDS rows per wave), 105 (register rows per slice) and 106 (bit 0 no streamed
rows, 1 no LDS rows, 2 no register rows, 3 no barrier wait, 4 no prefetch; bits 0-3 give meaningless numbers, only the
time counts) -> profiles/r02_persist_pcg_breakdown.txt
usage: ITERS=500 python tools/persist_breakdown.py c3d4|c3d10"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
wl = sys.argv[1]
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
print(wl, "maxrow", info.max_row_blocks, "nslices", info.nslices, flush=True)
for lds, rj, dbg in [(-1, 4, 0), (-1, 4, 16), (-1, 5, 0), (-1, 0, 0), (0, 0, 0), (-1, 4, 1), (-1, 4, 4), (-1, 4, 7), (-1, 4, 15)]:
    ctx.set_option(be.OPT_PCG_PERSIST, 1)
    ctx.set_option(106, dbg)
    ctx.set_option(104, lds)
    ctx.set_option(105, rj)
    for rep in range(2):
        t = time.perf_counter()
        try:
            it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=int(os.environ.get("ITERS", "200")))
            msg = f"it {it} rmax {rmax:.9e}"
        except Exception as e:
            msg = "EXC " + str(e)[:150]
        dt = time.perf_counter() - t
        nit = int(os.environ.get("ITERS", "200"))
        tm = ctx.timing()
        msg += f" paths 3k/small/persist {tm['solves_three']}/{tm['solves_small']}/{tm['solves_persist']}"
        print(f"  lds {lds} rj {rj} dbg {dbg} rep {rep}: {dt*1e3:8.2f} ms  {dt/nit*1e6:7.2f} us/it  {msg}", flush=True)
