"""us per PCG iteration on a shipped deck: one-launch small-system kernel vs the three-kernel loop (with / without graph)
usage: python tools/small_probe.py [deck=twist_plate_C3D10.inp]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from femcy_amd import backend as be
from femcy_amd.reader import InpInfo

name = sys.argv[1] if len(sys.argv) > 1 else "twist_plate_C3D10.inp"
inp = InpInfo(os.path.join(ROOT, "tests", "golden", "decks", name))
et = list(inp.eSets)[0]
ctx = be.Context(0)
ctx.set_mesh(inp.nodes, inp.eSets[et])
ctx.set_element(inp.ELE)
ctx.set_material(list(inp.materials.values())[0])
info = ctx.build_pattern()
dm = ctx.dm
ctx.assemble_K(-1)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * dm + b["dof"] for b in inp.dirichlet_bc_info]))
b = np.sin(np.arange(ctx.n) * 0.11) * 1e3
ctx.upload(be.VEC_RESIDUAL, b)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
print(f"{name}: n = {ctx.n}, nslices = {info.nslices}, nnzb = {info.nnzb}")
for label, small, graph, rr in (("one-launch", 1, 1, -1), ("one-launch, matrix streamed", 1, 1, 0),
                                ("3 kernels + graph", 0, 2, -1), ("3 kernels eager", 0, 0, -1)):
    ctx.set_option(be.OPT_PCG_SMALL, small)
    ctx.set_option(108, rr)
    ctx.set_option(be.OPT_PCG_GRAPH, graph)
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-12, maxit=10 * ctx.n)
    t = time.perf_counter()
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-12, maxit=10 * ctx.n)
    dt = time.perf_counter() - t
    x = ctx.download(be.VEC_X)
    print(f"  {label:<28} {it:6d} iterations, {dt*1e3:8.2f} ms, {dt/it*1e6:6.2f} us/iteration, rmax/r0 = {rmax/r0:.2e}, |x| = {np.linalg.norm(x):.12e}")
ctx.close()
