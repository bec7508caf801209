"""CPE8 / C3D10 / C3D4 product: wavefronts per slice (FEMCY_OPT_SPMV_VARIANT 1 / 2 / 4) launch to launch.
usage: python tools/r06_spmv_wps.py cpe8|c3d10|c3d4 [k] [quick]   (quick: a few products only, for a PMC pass)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from femcy_amd import backend as be, meshgen  # noqa: E402
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_quadrilateral, Element_quadratic_tetrahedral  # noqa: E402
from femcy_amd.material_zoo import LinearIsotropic, LinearIsotropicPlaneStrain  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cpe8"
if wl == "cpe8":
    kk = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1       # cpe8 2 = 2560 x 256 (3.9 M DOF, 1.1 GB)
    m = meshgen.beam_quad8(1280 * kk, 128 * kk, plane="CPE8")
    ele, mat = Element_quadratic_quadrilateral(), LinearIsotropicPlaneStrain(*m["elastic"])
else:
    k = int(sys.argv[2]) if len(sys.argv) > 2 else (6 if wl == "c3d10" else 12)
    m = meshgen.twist_plate_k(k, quadratic=wl == "c3d10")
    ele, mat = (Element_quadratic_tetrahedral() if wl == "c3d10" else Element_linear_tetrahedral()), LinearIsotropic(*m["elastic"])
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"])
ctx.set_element(ele)
ctx.set_material(mat)
info = ctx.build_pattern()
ctx.upload(be.VEC_DOF, np.zeros(ctx.n))
ctx.assemble_K(be.VEC_DOF)
spmv_b, iter_b = bench.algorithmic_bytes(info, ctx.nn, ctx.n)
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
if "quick" in sys.argv:
    print(f"{wl} {ctx.n} DOF: product {ctx.probe_spmv(5, True):.1f} us; algorithmic bytes {spmv_b}", flush=True)
    ctx.close()
    sys.exit(0)
if "ab" in sys.argv:                     # workgroups per XCD x round rotation, interleaved in ONE process (round 6)
    cfgs = [(256, 0), (512, 0), (256, 19), (512, 19), (256, 64), (512, 64), (0, -1)]     # (256, 0) = rounds 1-5, (0, -1) = the defaults
    if os.environ.get("FEMCY_AB_CFGS"):                           # e.g. "0:-1,256:0"
        cfgs = [tuple(int(v) for v in c.split(":")) for c in os.environ["FEMCY_AB_CFGS"].split(",")]
    for rep in range(3):
        for cap, rot in cfgs:
            ctx.set_option(be.TUNE_SPMV_WG_PER_XCD, cap)
            ctx.set_option(be.TUNE_SPMV_ROT, rot)
            us = sorted(ctx.probe_spmv(200, True) for _ in range(3))
            ctx.set_option(be.OPT_PCG_PERSIST, 0)
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20)
            ctx.set_option(be.OPT_TIMING, 64)
            ctx.timing_reset()
            its = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=200)[0]
            tm = ctx.timing()
            ctx.set_option(be.OPT_TIMING, 0)
            ctx.set_option(be.OPT_PCG_PERSIST, 1)
            print(f"{wl} {ctx.n} DOF rep {rep} wg/xcd {cap} rot {rot}: product {us[1]:.1f} us = {spmv_b / us[1] / 1e3 / 8000:.3f} of HBM; "
                  f"three-launch iteration {tm['pcg_ms'] * 1e3 / its:.1f} us = {iter_b / (tm['pcg_ms'] * 1e3 / its) / 1e3 / 8000:.3f}", flush=True)
    ctx.close()
    sys.exit(0)
for wps in (0, 1, 2, 4):
    ctx.set_option(be.OPT_SPMV_VARIANT, wps)
    us = min(ctx.probe_spmv(200, True) for _ in range(3))
    ctx.set_option(be.OPT_PCG_PERSIST, 0)
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=50)
    ctx.set_option(be.OPT_TIMING, 64)
    ctx.timing_reset()
    its = sum(ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=200)[0] for _ in range(3))
    tm = ctx.timing()
    ctx.set_option(be.OPT_TIMING, 0)
    ctx.set_option(be.OPT_PCG_PERSIST, 1)
    print(f"{wl} {ctx.n} DOF wps {wps}: product {us:.1f} us launch to launch = {spmv_b / us / 1e3 / 8000:.3f} of HBM; "
          f"three-launch iteration {tm['pcg_ms'] * 1e3 / its:.1f} us = {iter_b / (tm['pcg_ms'] * 1e3 / its) / 1e3 / 8000:.3f}", flush=True)
ctx.close()
