"""A/B of the PCG vector kernels' cache policy (default vs non-temporal, test knob 103) in one process.
usage: python tools/vec_policy_probe.py nx,ny,nz [rounds]"""
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
    ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
    for _ in range(3):
        ctx.pcg(be.VEC_TMP0, be.VEC_X, eps=0.0, maxit=500)
    for r in range(rounds):
        for nt in (0, 1):
            ctx.set_option(103, nt)
            ctx.pcg(be.VEC_TMP0, be.VEC_X, eps=0.0, maxit=100)
            t = time.perf_counter()
            it, _, _ = ctx.pcg(be.VEC_TMP0, be.VEC_X, eps=0.0, maxit=1000)
            print(f"  round {r} vector nt={nt}: {(time.perf_counter() - t) / it * 1e6:8.2f} us per PCG iteration")


if __name__ == "__main__":
    main()
