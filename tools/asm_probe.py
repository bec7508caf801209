"""time the assembly kernel alone (HIP events around the launch) on the C3D10 / C3D4 bench meshes.
usage: python tools/asm_probe.py c3d10|c3d4|cpe8 [mode=6] [reps=20]   (FEMCY_HIP_LIB selects an experimental build)
cpe8 = BASELINE configs[1]: the 1280 x 128 plane-strain beam of `bench.py --workload cpe8` (FEMCY_PROBE_CELLS=nx,ny)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral, Element_quadratic_quadrilateral
from femcy_amd.material_zoo import LinearIsotropic, LinearIsotropicPlaneStrain

wl = sys.argv[1] if len(sys.argv) > 1 else "c3d10"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else be.ASM_ROWS2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
quad = wl == "c3d10"
if wl == "cpe8":
    nx, ny = (int(v) for v in os.environ.get("FEMCY_PROBE_CELLS", "1280,128").split(","))
    m = meshgen.beam_quad8(nx, ny, plane="CPE8")
else:
    if os.environ.get("FEMCY_PROBE_K"):                         # BASELINE.md family: cells (8k, k, 12k)
        m = meshgen.twist_plate_k(int(os.environ["FEMCY_PROBE_K"]), quadratic=quad)
    else:
        m = (meshgen.twist_plate(48, 6, 72, quadratic=True, renumber=os.environ.get("FEMCY_BENCH_RENUM", "0") == "1") if quad
             else meshgen.twist_plate_k(12))
ctx = be.Context(0)
if os.environ.get("FEMCY_BENCH_SIGMA"):
    ctx.set_option(be.OPT_SELL_SIGMA, int(os.environ["FEMCY_BENCH_SIGMA"]))
if os.environ.get("FEMCY_PROBE_NODE_ORDER"):                    # FEMCY_OPT_NODE_ORDER: 0 caller's numbering, 1 measured choice
    ctx.set_option(be.OPT_NODE_ORDER, int(os.environ["FEMCY_PROBE_NODE_ORDER"]))
ctx.set_mesh(m["nodes"], m["elements"])
if wl == "cpe8":
    ctx.set_element(Element_quadratic_quadrilateral())
    ctx.set_material(LinearIsotropicPlaneStrain(*m["elastic"]))
else:
    ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
ctx.build_pattern()
ctx.set_option(be.OPT_ASSEMBLY, mode)
if os.environ.get("FEMCY_ROWS4_TILE"):                          # "GP,LCUT": FEMCY_TUNE_ROWS4_TILE = 1000 GP + LCUT
    gp, lcut = (int(v) for v in os.environ["FEMCY_ROWS4_TILE"].split(","))
    ctx.set_option(be.TUNE_ROWS4_TILE, 1000 * gp + lcut)
if os.environ.get("FEMCY_PROBE_ROWS4_ORDER"):                   # FEMCY_TUNE_ROWS4_ORDER: -1 auto, 0 longest first, 1 locality
    ctx.set_option(be.TUNE_ROWS4_ORDER, int(os.environ["FEMCY_PROBE_ROWS4_ORDER"]))
if os.environ.get("FEMCY_PROBE_PAIRS"):                         # FEMCY_TUNE_PAIRS bits
    ctx.set_option(be.TUNE_PAIRS, int(os.environ["FEMCY_PROBE_PAIRS"]))
ctx.upload(be.VEC_DOF, np.zeros(ctx.n))
for _ in range(3):
    ctx.assemble_K(be.VEC_DOF)
ctx.set_option(be.OPT_TIMING, 1)
ctx.timing_reset()
for _ in range(reps):
    ctx.assemble_K(be.VEC_DOF)
tm = ctx.timing()
info = ctx.pattern_info()
print(f"{wl}: {ctx.ne} elements, {ctx.n} DOF, nnzb {info.nnzb}, longest row {info.max_row_blocks} blocks")
print(f"{wl} mode {mode} knobs {os.environ.get('FEMCY_PROBE_PAIRS', '-')} rows4 order {os.environ.get('FEMCY_PROBE_ROWS4_ORDER', '-')} lib {os.path.basename(be.LIB_PATH)}: geom {tm['geom_ms']/tm['geom_launches']*1e3:.1f} us, "
      f"assemble {tm['assemble_ms']/tm['assemble_launches']*1e3:.1f} us")
ctx.close()
