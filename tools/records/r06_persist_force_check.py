"""FEMCY_DEBUG_FORCE_SPW=<n>: the persistent PCG of that shape against the three-launch loop on a small C3D10 plate"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
m = meshgen.twist_plate(24, 6, 36, quadratic=True)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"]); ctx.set_element(Element_quadratic_tetrahedral()); ctx.set_material(LinearIsotropic(*m["elastic"]))
ctx.build_pattern(); ctx.assemble_K(-1)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3); ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
out = {}
for persist in (0, 2):
    ctx.set_option(be.OPT_PCG_PERSIST, persist)
    out[persist] = [(ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k), ctx.download(be.VEC_X)) for k in (1, 9, 30)]
ok = all(a[0][0] == b[0][0] and abs(a[0][2] - b[0][2]) <= 1e-10 * a[0][2] and np.linalg.norm(a[1] - b[1]) <= 1e-10 * np.linalg.norm(a[1]) for a, b in zip(out[0], out[2]))
print("SPW", os.environ.get("FEMCY_DEBUG_FORCE_SPW"), "lib", os.path.basename(be.LIB_PATH), "paths", ctx.timing()["solves_persist"], "EQUAL" if ok else "DIFFERENT", [o[0] for o in out[2]], flush=True)
ctx.close()
