"""round 3: the persistent PCG's variants (FEMCY_TUNE_PERSIST_VARIANT bits: 1 non-temporal matrix stream, 2 tagged-granule
exchanges, 4 d in storage order with 16 + 8 byte gathers) on the headline mesh, and the ceilings the kernel runs
against (stream rate by buffer size and launch shape, exchange price by form).
usage: FEMCY_HIP_LIB=femcy_amd/libfemcy_hip_allvar.so ITERS=500 python tools/persist_variants.py [c3d4|c3d10] [variants...]
-> profiles/r03_persist_variants.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

wl = sys.argv[1] if len(sys.argv) > 1 else "c3d4"
variants = [int(v) for v in sys.argv[2:]] or list(range(8))
quad = wl == "c3d10"
nit = int(os.environ.get("ITERS", "500"))
m = meshgen.twist_plate(48, 6, 72, quadratic=True) if quad else meshgen.twist_plate_k(int(os.environ.get("K", "12")))
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"])
ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
ctx.set_material(LinearIsotropic(*m["elastic"]))
info = ctx.build_pattern()
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.assemble_K(-1)
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
streamed = ctx.persist_streamed_bytes()
print(f"{wl}: n {ctx.n} nslices {info.nslices} stored blocks {info.stored_blocks} "
      f"({info.stored_blocks * 76 / 1e6:.1f} MB), streamed per iteration {streamed / 1e6:.1f} MB", flush=True)

print("== ceilings: stream (GB/s) by buffer size; mode 0 = 1 WG/CU x 8 loads in flight, 1 = +nt, 2 = 8 WG/CU, 3 = +nt, "
      "4 / 5 = 1 WG/CU x 16 / 32 loads in flight, 6 / 7 = +nt", flush=True)
for mb in (16, 48, int(streamed / 2 ** 20) or 88, 112, 120, 128, 136, 144, 160, 200, 400, 1024):
    row = []
    for mode in range(8):
        best = 0.0
        for _ in range(2):
            g, moved = ctx.probe_stream(mb << 20, 20 if mb < 500 else 8, mode)
            best = max(best, g)
        row.append(best)
    print(f"  {mb:5d} MiB: " + "  ".join(f"m{k} {v:6.0f}" for k, v in enumerate(row)), flush=True)
print("== ceilings: grid-wide exchange (us) form 0 = counters + data, 1 = tagged granules", flush=True)
for form in (0, 1):
    try:
        v = [ctx.probe_exchange(2000, form) for _ in range(3)]
        print(f"  form {form}: " + " ".join(f"{x:.3f}" for x in v), flush=True)
    except be.FemcyError as e:
        print(f"  form {form}: FAILED {e}", flush=True)

ctx.set_option(be.OPT_PCG_PERSIST, 2 if quad else 1)
ref = None
runs = [(v, 1) for v in variants]
runs += [(v, l2) for v in variants if v & 1 for l2 in (0, 2, 3)]
for var, l2 in runs:
    try:
        ctx.set_option(be.TUNE_PERSIST_VARIANT, var)
        ctx.set_option(be.TUNE_PERSIST_L2_ROWS, l2)
        times = []
        for rep in range(4):
            t = time.perf_counter()
            it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
            times.append((time.perf_counter() - t) / nit * 1e6)
        x = ctx.download(be.VEC_X)
        tm = ctx.timing()
        if ref is None:
            ref = (x.copy(), rmax)
        dx = np.linalg.norm(x - ref[0]) / np.linalg.norm(ref[0])
        it2, _, rmax2 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
        same = np.array_equal(ctx.download(be.VEC_X), x)
        print(f"  variant {var} (nt {var & 1} a2a {(var >> 1) & 1} wide {(var >> 2) & 1}) l2rows {l2}: "
              + " ".join(f"{t:6.2f}" for t in times) + f" us/it  rmax {rmax:.6e} |x-x0|/|x0| {dx:.1e} "
              f"reproducible {same} paths 3k/small/persist {tm['solves_three']}/{tm['solves_small']}/{tm['solves_persist']}"
              f" timeouts {tm['barrier_timeouts']}", flush=True)
    except be.FemcyError as e:
        print(f"  variant {var}: FAILED {e}", flush=True)
