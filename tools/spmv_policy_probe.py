"""A/B of the SpMV matrix-stream cache policy (default vs non-temporal loads, test knob 102) in one process.
usage: python tools/spmv_policy_probe.py nx,ny,nz [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic


def main():
    cells = tuple(int(v) for v in sys.argv[1].split(","))
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    m = meshgen.twist_plate(*cells)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
    print(f"cells {cells}: {ctx.ne} elements, stored blocks {info.stored_blocks} = {info.stored_blocks * 76 / 2**20:.0f} MiB")
    reps = max(20, int(2e8 / ctx.ne))
    for r in range(rounds):
        for nt in (0, 1):
            ctx.set_option(102, nt)
            for _ in range(5):
                ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
            ctx.sync()
            t = time.perf_counter()
            for _ in range(reps):
                ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
            ctx.sync()
            dt = (time.perf_counter() - t) / reps
            t = time.perf_counter()
            it, _, _ = ctx.pcg(be.VEC_TMP0, be.VEC_X, eps=0.0, maxit=200)
            dp = (time.perf_counter() - t) / it
            print(f"  round {r} nt={nt}: {dt * 1e6:8.1f} us per SpMV back to back, {dp * 1e6:8.1f} us per PCG iteration")
        ctx.set_option(102, -1)
        t = time.perf_counter()
        it, _, _ = ctx.pcg(be.VEC_TMP0, be.VEC_X, eps=0.0, maxit=200)
        print(f"  round {r} auto: {(time.perf_counter() - t) / it * 1e6:8.1f} us per PCG iteration")


if __name__ == "__main__":
    main()
