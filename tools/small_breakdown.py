"""where the time of one iteration of the small-system one-launch PCG goes: fixed-length solves with parts of the
kernel switched off (probe build only: FEMCY_EXTRA_FLAGS=-DFEMCY_PERSIST_PROBE FEMCY_OUT=../libfemcy_hip_probe.so
csrc/build.sh; FEMCY_HIP_LIB=femcy_amd/libfemcy_hip_probe.so).  knob 106 bits: 32 no Ad loads, 64 no wait, 128 no Ad
stores, 256 no product, 512 no vector update; the numbers of such runs are meaningless, only the time counts.
usage: python tools/small_breakdown.py [deck]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from femcy_amd import backend as be
from femcy_amd.reader import InpInfo

name = sys.argv[1] if len(sys.argv) > 1 else "ellip_dense_CPS3_0d04.inp"
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
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
print(f"{name}: n = {ctx.n}, nslices = {info.nslices}, nnzb = {info.nnzb}")
NIT = 400
for dbg in (0, 32, 64, 128, 256, 512, 32 + 512, 64 + 128, 32 + 64 + 128, 32 + 64 + 128 + 256, 32 + 64 + 128 + 256 + 512):
    ctx.set_option(106, dbg)
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        try:
            it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=NIT)
        except Exception as e:
            it = -1
        best = min(best, time.perf_counter() - t)
    tm = ctx.timing()
    print(f"  dbg {dbg:4d}: it {it:5d}  {best/NIT*1e6:6.2f} us/iteration   small solves {tm['solves_small']}", flush=True)
ctx.close()
