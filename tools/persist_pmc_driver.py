"""a few solves of the headline system through k_pcg_persist, for rocprofv3 --pmc passes
usage: rocprofv3 --kernel-trace --pmc <counters> -d out -o pmc -- python tools/persist_pmc_driver.py [solves=3] [iters=200] [c3d4|c3d10|cpe8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = sys.argv[3] if len(sys.argv) > 3 else "c3d4"          # round 5: c3d10 (matrix streamed from HBM), cpe8 (2 x 2 blocks, 8 slices per wave)
ctx = be.Context(0)
if wl == "cpe8":
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    from femcy_amd.material_zoo import LinearIsotropicPlaneStrain
    m = meshgen.beam_quad8(1280, 128, plane="CPE8")
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_quadrilateral())
    ctx.set_material(LinearIsotropicPlaneStrain(*m["elastic"]))
else:
    from femcy_amd.element_zoo import Element_quadratic_tetrahedral
    m = meshgen.twist_plate(48, 6, 72, quadratic=True) if wl == "c3d10" else meshgen.twist_plate_k(12)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_tetrahedral() if wl == "c3d10" else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
ctx.build_pattern()
dm = m["nodes"].shape[1]
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * dm + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.assemble_K(-1)
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
for _ in range(solves):
    it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=iters)
print(f"{solves} solves of {iters} iterations, rmax {rmax:.6e}, paths {ctx.timing()['solves_persist']} persistent")
ctx.close()
