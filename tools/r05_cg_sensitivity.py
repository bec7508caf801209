"""How far does the reference's CG (eps = 1e-3 on max|r| / max|r0|, conjugateGradientSolver.py:103-127) pin its own
answer?  The C oracle (oracle/femcy_oracle.c: the reference's kernels as written) solves the SAME system with 1, 2, 4
and 8 OpenMP threads -- only the order of the floating-point sums in the four reductions changes, as it does from run to
run in Taichi (atomic adds) -- and reports the iteration count at the stop, and the distance of each solution from the
8-thread one and from the exact solution (sparse LU).  Systems: the first Newton solve of the twist plate k = 7
(116 280 DOF) after a twist increment of 0.003125 (no inverted element: K positive definite) and of 0.05 (state S1 of the
bench: the boundary layer is inverted, K indefinite).
usage: python tools/r05_cg_sensitivity.py            (parent: spawns one child per thread count)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(t1, out):
    from femcy_amd import meshgen
    from oracle import femcy_oracle as orc
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    m = meshgen.twist_plate_k(7)
    ed = elem_def("C3D4")
    topo = orc.Topology(m["nodes"], m["elements"], ed)
    mat = orc.Material("lin3d", m["elastic"])
    u = np.zeros(topo.n)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        orc.dirichlet_dof(u, bc, 3, topo.nodes, t1)
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
    cons = np.unique(np.concatenate(cons))
    co = COracle(m["nodes"], m["elements"], ed.dN_table(), ed.gauss_weights, mat.C, topo.adj_ptr, topo.adj_idx)
    f = co.internal_force(u, 0, *m["elastic"])
    co.get_dsdx_and_vol(u)
    co.assemble()
    co.zero_rows_cols_unit_diag(cons)
    f[cons] = 0.0
    x, it, r0, rmax = co.cg(f, eps=1e-3)
    np.savez(out, x=x, it=it, r0=r0, rmax=rmax, threads=co.threads())
    if os.environ.get("WITH_LU"):
        import scipy.sparse.linalg as sl
        np.save(out + ".lu.npy", sl.spsolve(co.to_csr().tocsc(), f))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(float(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    for t1, label in ((0.003125, "twist increment 0.003125 (K positive definite)"), (0.05, "twist increment 0.05 = state S1 (K indefinite)")):
        res = {}
        for th in (8, 4, 2, 1):
            out = f"/tmp/cg_sens_{t1}_{th}.npz"
            env = dict(os.environ, OMP_NUM_THREADS=str(th))
            if th == 8:
                env["WITH_LU"] = "1"
            subprocess.check_call([sys.executable, __file__, "child", str(t1), out], env=env)
            res[th] = np.load(out)
        xlu = np.load(f"/tmp/cg_sens_{t1}_8.npz.lu.npy")
        x8 = res[8]["x"]
        print(f"{label}: 116 280 DOF, max|r0| = {float(res[8]['r0']):.6e}")
        for th in (8, 4, 2, 1):
            x = res[th]["x"]
            print(f"  {th} threads: stops after {int(res[th]['it']):5d} iterations, max|r| / max|r0| = {float(res[th]['rmax']) / float(res[th]['r0']):.3e}, "
                  f"|x - x_8| / |x_8| = {np.linalg.norm(x - x8) / np.linalg.norm(x8):.3e}, |x - x_LU| / |x_LU| = {np.linalg.norm(x - xlu) / np.linalg.norm(xlu):.3e}",
                  flush=True)
