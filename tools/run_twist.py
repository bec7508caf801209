"""BASELINE configs[2] end to end: the synthetic twist plate through System_of_equations.solve
(increments + modified Newton + line searches, reference control flow) on one MI355X.
usage: python tools/run_twist.py [k=12] [max_time=1.0] [quadratic=0] [tangent=reference|consistent]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from femcy_amd import meshgen
from femcy_amd.body import Body
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
from femcy_amd.stiffnessMtrx import System_of_equations

k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
max_time = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
quad = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
tangent = sys.argv[4] if len(sys.argv) > 4 else "reference"
m = meshgen.twist_plate_k(k, quadratic=quad)
ELE = Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral()
ti = dict(m["time_incs"], max_time=max_time)
inp = SimpleNamespace(nodes=m["nodes"], eSets={m["etype"]: m["elements"]}, ELE=ELE,
                      dirichlet_bc_info=m["dirichlet_bc_info"], neumann_bc_info=[], time_incs=ti,
                      geometric_nonlinear=True, materials={"Elastic": LinearIsotropic(*m["elastic"])})
t0 = time.perf_counter()
body = Body(inp.nodes, m["elements"], ELE)
s = System_of_equations(body, inp.materials["Elastic"], True, verbose=False, tangent=tangent)
t1 = time.perf_counter()
s.solve(inp)
s.ctx.sync()
t2 = time.perf_counter()
u = s.dof.to_numpy()
print(f"k={k} quad={quad} tangent={tangent}: {m['elements'].shape[0]} elements, {u.size} DOF; setup {t1-t0:.2f} s, solve {t2-t1:.2f} s")
print(f"  increments {len(s.increments)} (failed {sum(not i['converged'] for i in s.increments)}), stats {s.stats}")
print(f"  |u|_2 = {np.linalg.norm(u):.6f}, max|u| = {np.abs(u).max():.6f}, time reached {s.time0}")
