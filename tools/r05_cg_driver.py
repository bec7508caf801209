"""the reference's CG branch (>= 1e5 DOF: solve_by_CG, eps = 1e-3, maxit = n; stiffnessMtrx.py:254-276) through the
product driver on a generated twist plate: per-solve iteration counts, increments, wall time.
usage: python tools/r05_cg_driver.py [k=7] [max_time=0.05] [ini_inc]"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import meshgen
from femcy_amd.body import Body
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
from femcy_amd.stiffnessMtrx import System_of_equations

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
m = meshgen.twist_plate_k(k)
ti = dict(m["time_incs"])
ti["max_time"] = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
if len(sys.argv) > 3:
    ti["ini_inc"] = float(sys.argv[3])
ELE = Element_linear_tetrahedral()
inp = SimpleNamespace(nodes=m["nodes"], eSets={"C3D4": m["elements"]}, ELE=ELE, dirichlet_bc_info=m["dirichlet_bc_info"],
                      neumann_bc_info=[], time_incs=ti, geometric_nonlinear=True,
                      materials={"Elastic": LinearIsotropic(*m["elastic"])})
s = System_of_equations(Body(inp.nodes, m["elements"], ELE), inp.materials["Elastic"], True, verbose=False)
t = time.time()
s.solve(inp)
s.ctx.sync()
dt = time.time() - t
print(f"k {k}: {s.ctx.ne} elements, {s.ctx.n} DOF, max_time {ti['max_time']}: {dt:.1f} s, stats {s.stats}")
print("increments:", [(round(i['time1'], 6), i['converged'], i['newton_loop']) for i in s.increments])
print("cg:", [(c['iters'], "%.3e" % c['r0'], "%.3e" % c['rmax']) for c in s.cg_log])
u = s.dof.to_numpy()
print("|u| = %.12e  max|u| = %.12e" % (np.linalg.norm(u), np.abs(u).max()))
if os.environ.get("SAVE"):
    np.savez_compressed(os.environ["SAVE"], dof=u, cg=np.array([(c['iters'], c['r0'], c['rmax'], c['time1']) for c in s.cg_log]),
                        inc=np.array([(i['time1'], i['dt'], i['converged'], i['newton_loop']) for i in s.increments], dtype=float))
