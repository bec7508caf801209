"""debug: the persistent PCG on a C3D10 plate of given cells, register rows on / off; prints path and time"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
cells = tuple(int(v) for v in sys.argv[1].split(","))
rj = int(sys.argv[2]) if len(sys.argv) > 2 else -1
m = meshgen.twist_plate(*cells, quadratic=True)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"]); ctx.set_element(Element_quadratic_tetrahedral()); ctx.set_material(LinearIsotropic(*m["elastic"]))
info = ctx.build_pattern()
ctx.assemble_K(-1)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
if rj >= 0:
    ctx.set_option(105, rj)
if os.environ.get("PDBG"):
    ctx.set_option(106, int(os.environ["PDBG"]))
print(cells, "slices", info.nslices, "rj", rj, "pdbg", os.environ.get("PDBG"), flush=True)
r = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=5)
print("solve ok", r, ctx.timing()["solves_persist"], flush=True)
ctx.close()
