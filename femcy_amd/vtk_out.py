"""legacy-VTK writer: the headless replacement for the reference's GGUI windows (`Body.show`,
body.py:100-162; the reference README lists file output as future work).  Writes the mesh, the nodal
displacement and, per cell, the Gauss-point mean of the von Mises stress."""
import numpy as np

# VTK cell type and the permutation from the reference's local node order to VTK's
_VTK = {("tri", 3): (5, [0, 1, 2]), ("tri", 6): (22, [0, 1, 2, 3, 4, 5]),
        ("quad", 4): (9, [0, 1, 2, 3]), ("quad", 8): (23, [0, 1, 2, 3, 4, 5, 6, 7]),
        # reference tets: node0 at zeta=1, node1 at xi=1, node2 origin, node3 at eta=1; mid-sides 4:(0,1)
        # 5:(1,2) 6:(2,0) 7:(0,3) 8:(3,1) 9:(2,3).  VTK quadratic tet edges: (0,1)(1,2)(2,0)(0,3)(1,3)(2,3)
        ("tet", 4): (10, [0, 1, 2, 3]), ("tet", 10): (24, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])}


def write_vtk(path: str, system, title: str = "femcy_amd result"):
    nodes = np.asarray(system.body.np_nodes)
    el = np.asarray(system.body.np_elements)
    dm, npe = nodes.shape[1], el.shape[1]
    family = "tet" if dm == 3 else ("tri" if npe in (3, 6) else "quad")
    ctype, perm = _VTK[(family, npe)]
    u = system.dof.to_numpy().reshape(-1, dm)
    pts = np.zeros((nodes.shape[0], 3))
    pts[:, :dm] = nodes
    disp = np.zeros_like(pts)
    disp[:, :dm] = u
    try:
        mises = system.mises_stress.to_numpy().mean(axis=1)
    except Exception:
        mises = None
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 3.0\n%s\nASCII\nDATASET UNSTRUCTURED_GRID\n" % title)
        f.write("POINTS %d double\n" % pts.shape[0])
        np.savetxt(f, pts, fmt="%.10g")
        f.write("CELLS %d %d\n" % (el.shape[0], el.shape[0] * (npe + 1)))
        np.savetxt(f, np.concatenate([np.full((el.shape[0], 1), npe), el[:, perm]], axis=1), fmt="%d")
        f.write("CELL_TYPES %d\n" % el.shape[0])
        np.savetxt(f, np.full(el.shape[0], ctype), fmt="%d")
        f.write("POINT_DATA %d\nVECTORS displacement double\n" % pts.shape[0])
        np.savetxt(f, disp, fmt="%.10g")
        if mises is not None:
            f.write("CELL_DATA %d\nSCALARS mises double 1\nLOOKUP_TABLE default\n" % el.shape[0])
            np.savetxt(f, mises, fmt="%.10g")
