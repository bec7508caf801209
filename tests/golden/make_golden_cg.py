"""Golden vectors of the reference's CG BRANCH (`solve_dof` at >= 1e5 DOF -> `solve_by_CG`: Jacobi-PCG, eps = 1e-3,
maxit = n; /root/reference/stiffnessMtrx.py:254-276, conjugateGradientSolver.py:103-127) driven through the
increment / Newton / line-search drivers (`solve`, `advance_inc`, :647-822) -- produced by the CPU oracle
(`OracleSystem(cg_backend="c")`: numpy assembly and residuals, the as-written C CG of oracle/femcy_oracle.c).

Like tests/golden/make_golden.py these are ORACLE outputs, not outputs of the reference (Taichi cannot run here).
Every shipped deck is below 1e5 DOF and takes the spsolve branch, so the systems are generated (femcy_amd.meshgen):

  twist_k7      twist_plate_k(7): 197 568 C3D4, 116 280 DOF, nlgeom, user Dirichlet BC, to max_time = 0.05 with the
                deck's own *Static line (0.05, 1, 1e-5, 0.05).  The first increment asks for 9 degrees of twist in one
                step on 1.43-unit elements: the boundary layer inverts, K = sum B^T C B det(J) w goes indefinite, CG
                runs to its cap maxit = n = 116 280 without converging, the Newton residual explodes, the increment
                is cut back twice -- that IS the reference's behaviour on this system (its CG has no breakdown test)
  twist_k7_fine the same mesh with ini_inc = max_inc = 0.003125 (what the cut-backs arrive at), to max_time = 0.0125:
                every solve converges -- the clean comparison of per-solve iteration counts
  twist_c3d10_fine  BASELINE configs[4]'s element on the CG branch: 36 864 C3D10 (32 x 4 x 48 cells), 170 235 DOF, two
                increments of 0.003125
  beam_lin      linear (nlgeom = NO) CPE8 cantilever 420 x 42 serendipity quadrilaterals, 107 690 DOF, tip displacement
                1: ONE CG solve at eps = 1e-3

For each: the increment list, one row per CG solve (iterations, max|r0|, max|r|, time1), the final dof, and the final
dof of the SAME system with every linear solve exact (`linear_solver="spsolve"`: the eps -> 0 yardstick of SURVEY 7).

    python tests/golden/make_golden_cg.py [case ...]      # ~25 min for twist_k7 on 8 cores
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from femcy_amd import meshgen  # noqa: E402
from oracle.femcy_oracle import Material, OracleSystem  # noqa: E402


def cases():
    tw = meshgen.twist_plate_k(7)
    ti = dict(tw["time_incs"], max_time=0.05)
    fine = dict(tw["time_incs"], ini_inc=0.003125, max_inc=0.003125, max_time=0.0125)
    bm = beam_lin_mesh()
    tq = meshgen.twist_plate(32, 4, 48, quadratic=True)
    fineq = dict(tq["time_incs"], ini_inc=0.003125, max_inc=0.003125, max_time=0.00625)
    return {
        "twist_c3d10_fine": (tq, "C3D10", Material("lin3d", tq["elastic"]), True, fineq),
        "twist_k7": (tw, "C3D4", Material("lin3d", tw["elastic"]), True, ti),
        "twist_k7_fine": (tw, "C3D4", Material("lin3d", tw["elastic"]), True, fine),
        "beam_lin": (bm, "CPE8", Material("pstrain", bm["elastic"]), False, bm["time_incs"]),
    }


def beam_lin_mesh():
    m = meshgen.beam_quad8(420, 42, plane="CPE8", tip_disp=1.0)
    m["geometric_nonlinear"] = False
    m["time_incs"] = {"ini_inc": 1.0, "max_time": 1.0, "min_inc": 1e-5, "max_inc": 1.0}
    return m


def run(name, mesh, etype, mat, nlgeom, ti, solver):
    s = OracleSystem(mesh["nodes"], mesh["elements"], etype, mat, nlgeom, linear_solver=solver, cg_backend="c",
                     verbose="-v" in sys.argv)
    t = time.time()
    u = s.solve(ti, mesh["dirichlet_bc_info"], mesh["neumann_bc_info"])
    cg = np.array([(l["iters"], l["r0"], l["rmax"], l["time1"]) for l in s.log if l["solve"] == "cg"], dtype=np.float64)
    inc = np.array([(i["time1"], i["dt"], float(i["converged"]), i["newton_loop"]) for i in s.increments], dtype=np.float64)
    print(f"{name} [{solver}]: {len(s.increments)} increments, {s.n_solves} solves, "
          f"{int(cg[:, 0].sum()) if cg.size else 0} CG iterations, |u| = {np.linalg.norm(u):.10g} ({time.time() - t:.0f} s)", flush=True)
    return u, cg.reshape(-1, 4), inc, s


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("-")]
    path = os.path.join(HERE, "oracle_cg_branch.npz")
    out = {}
    if os.path.exists(path):
        old = np.load(path)
        out = {k: old[k] for k in old.files}
    for name, (mesh, etype, mat, nlgeom, ti) in cases().items():
        if want and name not in want:
            continue
        u, cg, inc, s = run(name, mesh, etype, mat, nlgeom, ti, "reference")
        out[name + "/dof"], out[name + "/cg"], out[name + "/inc"] = u, cg, inc
        out[name + "/meta"] = np.array([s.n_solves, s.n_assemblies, getattr(s, "ini_residual", 0.0), s.time0], dtype=np.float64)
        # (twist_k7: the exact-solve flow is twist_k7_fine's; twist_c3d10_fine: a sparse LU of 170 k C3D10 unknowns per
        # solve is hours of SuperLU)
        if "--no-tight" not in sys.argv and name not in ("twist_k7", "twist_c3d10_fine"):
            ut, _, inct, st = run(name, mesh, etype, mat, nlgeom, ti, "spsolve")
            out[name + "/dof_tight"], out[name + "/inc_tight"] = ut, inct
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
