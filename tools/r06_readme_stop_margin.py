"""README.md:70 (CPS6 93.32 / 84.40 = iterate 128 of the reference's CG at eps = 1e-3): how wide is the stop test's margin?
For every assembly variant of the 2-D families: iterations at the stop, max|r| / (eps max|r0|) at iterates 126..130 (forced
by maxit), sigma_yy at D.  usage: python tools/r06_readme_stop_margin.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import deck  # noqa: E402
from femcy_amd import backend as be  # noqa: E402
from femcy_amd.body import Body  # noqa: E402
from femcy_amd.reader import InpInfo  # noqa: E402
from femcy_amd.stiffnessMtrx import System_of_equations  # noqa: E402

names = {be.ASM_GATHER: "gather", be.ASM_GATHER_SYM: "gather_sym", be.ASM_GATHER_SYM_ROWSUM: "gather_sym_rowsum",
         be.ASM_ATOMIC: "atomic", be.ASM_ROWS: "rows", be.ASM_PAIRS: "pairs"}
for dk, want in (("ellip_membrane_quadritic_trig_neumann.inp", 128), ("ellip_membrane_linEle_localVeryFine.inp", 105)):
    for mode, name in names.items():
        inp = InpInfo(deck(dk))
        body = Body(nodes=inp.nodes, elements=list(inp.eSets.values())[0], ELE=inp.ELE)
        s = System_of_equations(body, list(inp.materials.values())[0], False, verbose=False, direct="pcg", direct_eps=1e-3)
        s.ctx.set_option(be.OPT_ASSEMBLY, mode)
        s.solve(inp)
        it = s.PCG.iterations
        s.compute_strain_stress()
        syy = s.cauchy_stress.to_numpy()[:, :, 1, 1].max()
        line = f"{dk[:34]:34s} {name:18s} stop at {it:4d} (rmax / (eps r0) = {s.PCG.rmax / (1e-3 * s.PCG.r0):.6f}) max sigma_yy {syy:.4f} | "
        for k in range(want - 2, want + 3):
            i2, r0, rmax = s.ctx.pcg(s.PCG.b.id, s.PCG.x.id, eps=0.0, maxit=k)
            line += f"{k}: {rmax / (1e-3 * r0):.5f}  "
        print(line, flush=True)
        s.ctx.close()
