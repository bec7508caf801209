"""persistent one-launch PCG vs three launches per iteration over the mesh size (C3D4 twist plates), us per iteration
usage: python tools/persist_size_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

ITERS = 300
print(f"{'cells':>14} {'elements':>9} {'nodes':>8} {'slices':>7} {'K MB':>6} | {'3 launches':>10} {'persistent':>10}  path")
for cells in ((24, 6, 96), (32, 6, 96), (40, 8, 96), (48, 12, 96), (64, 12, 120), (80, 12, 144), (96, 12, 144), (100, 12, 152), (100, 13, 160), (104, 14, 160)):
    m = meshgen.twist_plate(*cells)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern()
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    us = {}
    for persist in (0, 1, 2):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=50)
        before = ctx.timing()["solves_persist"]
        t = time.perf_counter()
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=ITERS)
        us[persist] = (time.perf_counter() - t) / ITERS * 1e6
        took = ctx.timing()["solves_persist"] - before
        if persist == 1:
            took1 = took
    kmb = info.stored_blocks * 76 / 1e6
    print(f"{'x'.join(map(str, cells)):>14} {m['elements'].shape[0]:>9} {m['nodes'].shape[0]:>8} {info.nslices:>7} {kmb:>6.0f} | "
          f"{us[0]:>10.1f} {us[1]:>10.1f}  {'persistent' if took1 else 'three launches (not eligible)'}"
          f"{'' if took1 else f'; forced persistent {us[2]:.1f}'}", flush=True)
    ctx.close()
