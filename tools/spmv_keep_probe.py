"""NT SpMV with a share of the matrix kept on the default cache policy (knob 110): us per PCG iteration
usage: python tools/spmv_keep_probe.py c3d10|c3d4_8m|c3d4_2m"""
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
cells = {"c3d10": (48, 6, 72), "c3d4_8m": (192, 24, 288), "c3d4_2m": (120, 16, 176)}[wl]
m = meshgen.twist_plate(*cells, quadratic=quad)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"])
ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
ctx.set_material(LinearIsotropic(*m["elastic"]))
info = ctx.build_pattern()
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.assemble_K(-1)
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
ctx.set_option(be.OPT_PCG_PERSIST, 0)
kmb = info.stored_blocks * 76 / 1e6
print(f"{wl}: {m['elements'].shape[0]} elements, stored matrix {kmb:.0f} MB", flush=True)
nit = 200
for keep in ((0, -1, 100, 140, 180, 0, -1) if wl == 'c3d4_8m' else (0, -1, 520, 580, 620, 680, 720, 0, -1)):
    ctx.set_option(110, keep)
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
    best = 1e9
    for rep in range(2):
        t = time.perf_counter()
        it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
        best = min(best, (time.perf_counter() - t) / nit * 1e6)
    print(f"  keep {keep:4d} per mille ({(kmb * keep / 1000 if keep >= 0 else 220):6.0f} MB cacheable): {best:7.2f} us / iteration   rmax {rmax:.6e}", flush=True)
