"""femcy_direct_solve against the tight PCG it replaced, per linear solve on the systems of a few decks (ms, median of
`reps` solves after one warm-up; the factorisation is redone on every call, as in a Newton iteration).
usage: python tools/direct_bench.py [deck ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from femcy_amd import backend as be                      # noqa: E402
from femcy_amd.reader import InpInfo                     # noqa: E402

DECKS = ["beamFreeDeflect_CPS6_load_mesh13", "gen_beam_CPE8_tip4", "twist_plate_C3D4", "twist_C3D10_coarse",
         "cookMembrane_CPE6_smallDef", "cookMembrane_CPE6_smallDef_nu0d4999", "twist_plate_C3D10", "ellip_dense_CPS6_0d04"]


def main():
    names = sys.argv[1:] or DECKS
    reps = int(os.environ.get("REPS", "5"))
    for name in names:
        inp = InpInfo(os.path.join(ROOT, "tests", "golden", "decks", name + ".inp"))
        el = list(inp.eSets.values())[0]
        ctx = be.Context(0)
        ctx.set_mesh(inp.nodes, el)
        ctx.set_element(inp.ELE)
        ctx.set_material(list(inp.materials.values())[0])
        ctx.build_pattern()
        if os.environ.get("VARIANT"):                          # FEMCY_TUNE_DIRECT_UPDATE: 0 VALU, 1 / 2 matrix cores
            ctx.set_option(be.TUNE_DIRECT_UPDATE, int(os.environ["VARIANT"]))
        ctx.assemble_K(-1)
        cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * ctx.dm + b["dof"] for b in inp.dirichlet_bc_info]))
        ctx.upload(be.VEC_RESIDUAL, np.random.default_rng(0).standard_normal(ctx.n))
        ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)

        def timed(fn):
            fn()
            ts = []
            for _ in range(reps):
                ctx.sync()
                t0 = time.perf_counter()
                out = fn()
                ctx.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts)), out
        td, info = timed(lambda: ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X))
        xd = ctx.download(be.VEC_X)
        tp, res = timed(lambda: ctx.pcg(be.VEC_RESIDUAL, be.VEC_TMP0, eps=1e-12, maxit=10 * ctx.n))
        xp = ctx.download(be.VEC_TMP0)
        print(f"{name:42s} n {ctx.n:6d}  band {info['bandwidth']:5d} ({info['band_bytes'] / 1e6:7.1f} MB, {info['panels']:4d} panels)  "
              f"direct {td:8.2f} ms (residual {info['residual']:.1e}, {info['refinements']} refinements)   tight PCG {tp:8.2f} ms "
              f"({res[0]} iterations, max|r|/max|r0| {res[2] / res[1]:.1e})   |x_d - x_p| / |x_d| {np.linalg.norm(xd - xp) / np.linalg.norm(xd):.1e}",
              flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
