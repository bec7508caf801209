"""femcy_direct_solve towards the limit of the reference's direct branch (1e5 DOF): cube-like 3-D meshes, whose band is
the widest a mesh of that size can have.  Prints band size, time per solve, residual, and the tight PCG beside it.
usage: python tools/direct_limit.py [cells ...]   (default 12 20 30: 6.6 k / 27.8 k / 89.4 k DOF)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from femcy_amd import backend as be, meshgen                          # noqa: E402
from femcy_amd.element_zoo import Element_linear_tetrahedral         # noqa: E402
from femcy_amd.material_zoo import LinearIsotropic                   # noqa: E402


def main():
    for k in [int(a) for a in sys.argv[1:]] or [12, 20, 30]:
        m = meshgen.twist_plate(k, k, k)
        ctx = be.Context(0)
        ctx.set_mesh(m["nodes"], m["elements"])
        ctx.set_element(Element_linear_tetrahedral())
        ctx.set_material(LinearIsotropic(*m["elastic"]))
        ctx.build_pattern()
        if os.environ.get("VARIANT"):                          # FEMCY_TUNE_DIRECT_UPDATE: 0 VALU, 1 / 2 matrix cores
            ctx.set_option(be.TUNE_DIRECT_UPDATE, int(os.environ["VARIANT"]))
        ctx.assemble_K(-1)
        cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
        ctx.upload(be.VEC_RESIDUAL, np.random.default_rng(0).standard_normal(ctx.n))
        ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
        b = ctx.download(be.VEC_RESIDUAL)
        t0 = time.perf_counter()
        info = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)          # first call: reverse Cuthill-McKee + allocations
        t_first = (time.perf_counter() - t0) * 1e3
        ts = []
        for _ in range(5):                                           # median of five
            ctx.sync()
            t0 = time.perf_counter()
            info = ctx.direct_solve(be.VEC_RESIDUAL, be.VEC_X)
            ctx.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        t_d = sorted(ts)[2]
        x = ctx.download(be.VEC_X)
        K = ctx.get_K_bsr().tocsr()
        res = np.abs(K @ x - b).max() / np.abs(b).max()
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_TMP0, eps=1e-12, maxit=10 * ctx.n)
        tp = []
        for _ in range(5):
            ctx.sync()
            t0 = time.perf_counter()
            it = ctx.pcg(be.VEC_RESIDUAL, be.VEC_TMP0, eps=1e-12, maxit=10 * ctx.n)
            ctx.sync()
            tp.append((time.perf_counter() - t0) * 1e3)
        t_p = sorted(tp)[2]
        xp = ctx.download(be.VEC_TMP0)
        print(f"{k}^3 cells: {ctx.n} DOF, {info['bandwidth']} sub-diagonals, {info['panels']} panels, band {info['band_bytes'] / 1e9:.2f} GB: "
              f"direct {t_d:.1f} ms (median of 5: {' '.join('%.1f' % v for v in ts)}; first call {t_first:.0f} ms), residual {res:.1e} ({info['refinements']} refinements); tight PCG "
              f"{t_p:.1f} ms ({it[0]} iterations); |x_d - x_p| / |x_d| = {np.linalg.norm(x - xp) / np.linalg.norm(x):.1e}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
