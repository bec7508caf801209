"""Small driver for rocprofv3 passes: a few launches of every hot-path kernel on one of the bench meshes.
usage: python tools/prof_workload.py c3d4|c3d10|c3d4_8m [asm_reps] [spmv_reps] [pcg_iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
from femcy_amd.user_defined import user_dirichletBC_values


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3d10"
    asm_reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    spmv_reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    pcg_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    quad = wl == "c3d10"
    m = meshgen.twist_plate(48, 6, 72, quadratic=True) if quad else meshgen.twist_plate_k(24 if wl == "c3d4_8m" else 12)
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    for k, v in (("FEMCY_PROF_ASSEMBLY", be.OPT_ASSEMBLY), ("FEMCY_PROF_SPMV_VARIANT", be.OPT_SPMV_VARIANT),
                 ("FEMCY_PROF_STORAGE_ORDER", be.OPT_PCG_STORAGE_ORDER), ("FEMCY_PROF_PERSIST", be.OPT_PCG_PERSIST)):
        if os.environ.get(k):
            ctx.set_option(v, int(os.environ[k]))
    ctx.build_pattern()
    u = np.zeros(ctx.n)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
        if bc["user"]:
            user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
    cons = np.unique(np.concatenate(cons))
    ctx.upload(be.VEC_DOF, u)
    for _ in range(asm_reps):
        ctx.assemble_K(be.VEC_DOF)
        ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    ctx.upload(be.VEC_TMP0, np.random.default_rng(0).standard_normal(ctx.n))
    for _ in range(spmv_reps):
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=pcg_iters)
    ctx.sync()
    ctx.close()


if __name__ == "__main__":
    main()
