"""why did the three-launch PCG on the C3D10 plate take 204-235 us / iteration inside tests/test_gpu_pcg_persist.py (78 in
bench.py / tools/r05_ab.py)?  The test's sequence, with the product timed by HIP events (OPT_TIMING) and by wall clock,
before and after persistent solves, with and without the hipGraph replay."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

m = meshgen.twist_plate(48, 6, 72, quadratic=True)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"])
ctx.set_element(Element_quadratic_tetrahedral())
ctx.set_material(LinearIsotropic(*m["elastic"]))
ctx.build_pattern()
mode = sys.argv[1] if len(sys.argv) > 1 else "zero"
if mode == "zero":
    ctx.assemble_K(-1)
else:
    u = np.zeros(ctx.n)
    u[::7] = 1e-3
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
b = np.sin(np.arange(ctx.n) * 0.11) * 1e3
ctx.upload(be.VEC_RESIDUAL, b)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)


def wall(maxit):
    ctx.sync()
    t = time.perf_counter()
    r = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=maxit)
    ctx.sync()
    return (time.perf_counter() - t) / maxit * 1e6, r


def events(maxit):
    ctx.set_option(be.OPT_TIMING, 1)
    ctx.timing_reset()
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=maxit)
    tm = ctx.timing()
    ctx.set_option(be.OPT_TIMING, 0)
    return tm["spmv_ms"] * 1e3 / max(tm["spmv_launches"], 1), tm["pcg_ms"] * 1e3 / maxit


for label, opts in (("PERSIST=0", [(be.OPT_PCG_PERSIST, 0)]), ("PERSIST=1 MAX_MB=240", [(be.OPT_PCG_PERSIST, 1), (be.TUNE_PERSIST_MAX_MB, 240)]),
                    ("persistent (MAX_MB=0)", [(be.TUNE_PERSIST_MAX_MB, 0)]),
                    ("PERSIST=1 MAX_MB=240 again", [(be.TUNE_PERSIST_MAX_MB, 240)]), ("PERSIST=0 again", [(be.OPT_PCG_PERSIST, 0)]),
                    ("PERSIST=0, no graph", [(be.OPT_PCG_GRAPH, 0)])):
    for o, v in opts:
        ctx.set_option(o, v)
    w100, r = wall(100)
    w300 = [wall(300)[0] for _ in range(3)]
    w20, r20 = wall(20)
    ev = events(96)
    tm = ctx.timing()
    print(f"{mode} {label:28s}: wall us/it  100: {w100:7.1f}  300: " + " ".join(f"{v:6.1f}" for v in w300) + f"  20: {w20:7.1f} | events: product {ev[0]:6.1f} us, "
          f"iteration {ev[1]:6.1f} us | r(20 its) {r20} | three/persist so far {tm['solves_three']}/{tm['solves_persist']}", flush=True)
ctx.close()
