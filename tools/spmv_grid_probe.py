"""sweep of the SpMV workgroups-per-XCD cap (test knob 101) and wavefronts per slice, back-to-back SpMV timing.
usage: python tools/spmv_grid_probe.py k quadratic(0/1)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic


def main():
    k, quad = int(sys.argv[1]), bool(int(sys.argv[2]))
    m = meshgen.twist_plate_k(k, quadratic=quad)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
    print(f"k={k} quad={quad}: nslices {info.nslices}, stored blocks {info.stored_blocks}")
    reps = 200
    for wps in (1, 2, 4):
        ctx.set_option(be.OPT_SPMV_VARIANT, wps)
        row = []
        for cap in (24, 32, 48, 64, 96, 128, 160, 192, 256, 384, 512):
            ctx.set_option(101, cap)
            for _ in range(5):
                ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
            ctx.sync()
            t = time.perf_counter()
            for _ in range(reps):
                ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
            ctx.sync()
            row.append(f"{cap}:{(time.perf_counter() - t) / reps * 1e6:.1f}")
        print(f"  wps={wps}  " + "  ".join(row))


if __name__ == "__main__":
    main()
