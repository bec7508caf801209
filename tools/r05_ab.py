"""round 5 A/B runs on one GPU (one process per library build; FEMCY_HIP_LIB selects the build):
  persist_hbm   the persistent PCG FORCED (FEMCY_OPT_PCG_PERSIST = 2) on the 124 k C3D10 plate, whose matrix (380 MB)
                streams from HBM -- the case the 240 MiB rule (ctx.hpp) bars -- against the three-launch loop:
                register rows x LDS rows x default-policy rows; iterates compared with the three-launch loop
usage: [FEMCY_HIP_LIB=...] python tools/r05_ab.py persist_hbm [combos]   combos = "rj:lds:l2rows:dbg,..." """
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be
from r04_ab import problem, make_ctx, state

HBM = 8000.0


def persist_hbm(combos=None, wl="c3d10"):
    m, quad, u, cons = problem(wl)
    nit = int(os.environ.get("ITERS", "300"))
    ctx, info = make_ctx(m, quad, [])
    state(ctx, u, cons)
    spmv_b = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n
    iter_b = spmv_b + 88 * ctx.n
    print(f"lib {os.path.basename(be.LIB_PATH)}  {wl}: n {ctx.n}, slices {info.nslices if hasattr(info, 'nslices') else '?'}, "
          f"8d bytes / iteration {iter_b / 1e6:.1f} MB", flush=True)

    def run(label):
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
        times = []
        for _ in range(4):
            ctx.sync()
            t = time.perf_counter()
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
            times.append((time.perf_counter() - t) / nit * 1e6)
        r30 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        x30 = ctx.download(be.VEC_X)
        tm = ctx.timing()
        best = min(times)
        return times, best, r30, x30, tm

    ctx.set_option(be.OPT_PCG_PERSIST, 0)
    times, best, r30, xref, tm = run("three")
    print(f"  three launches            : " + " ".join(f"{t:6.2f}" for t in times) + f" us/it  = {iter_b / best / 1e3 / HBM:.3f} of HBM "
          f"| 30 its rmax {r30[2]:.9e}", flush=True)
    ctx.set_option(be.OPT_PCG_PERSIST, 2)
    if combos is None:
        combos = "4:-1:1:0,4:-1:0:0,4:-1:4:0,5:-1:1:0,0:-1:1:0,4:0:1:0,4:-1:1:16"
    for cb in combos.split(","):
        rj, lds, l2, dbg = (int(v) for v in cb.split(":"))
        try:
            ctx.set_option(105, rj)
            ctx.set_option(104, lds)
            ctx.set_option(be.TUNE_PERSIST_L2_ROWS, l2)
            ctx.set_option(106, dbg)
            before = ctx.timing()
            times, best, r30, x30, tm = run(cb)
            d30 = np.linalg.norm(x30 - xref) / np.linalg.norm(xref)
            print(f"  persist rj {rj} lds {lds:2d} l2rows {l2} dbg {dbg:2d}: " + " ".join(f"{t:6.2f}" for t in times) +
                  f" us/it  = {iter_b / best / 1e3 / HBM:.3f} of HBM | streamed {ctx.persist_streamed_bytes() / 1e6:.1f} MB "
                  f"| 30 its rmax {r30[2]:.9e} |x-x3|/|x3| {d30:.1e} | persist/three/timeouts "
                  f"{tm['solves_persist'] - before['solves_persist']}/{tm['solves_three'] - before['solves_three']}/{tm['barrier_timeouts']}", flush=True)
        except be.FemcyError as e:
            print(f"  persist {cb}: FAILED {e}", flush=True)
    ctx.close()


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "persist_hbm":
        persist_hbm(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None, sys.argv[3] if len(sys.argv) > 3 else "c3d10")
    else:
        raise SystemExit(__doc__)
