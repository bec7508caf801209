"""Deterministic synthetic twist-plate meshes (no RNG) for the benchmark / scaling configs.

The shipped twist decks have ~1e3 elements (tests/golden/decks/twist_plate_C3D4.inp: plate
80 x 10 x 120, face z=120 clamped = `Set-10`, face z=0 rotated by time*pi about (40,5) =
`fit_right_z` through `*Boundary, user`, `*Elastic 2e11, 0.3`, nlgeom=YES,
`*Static 0.05,1,1e-5,0.05`).  The generator reproduces that model on a structured grid of
(nx, ny, nz) hexahedral cells, each split into 6 Kuhn tetrahedra (conforming without parity
tricks).  Local node order is chosen so that det(J) > 0 under the reference's C3D4 map
N = [zeta, xi, 1-xi-eta-zeta, eta] (element_linear_tetrahedral.py:68-82).

BASELINE.md sizes: k=12 -> cells 96x12x144 -> 995 328 C3D4, 182 845 nodes, 548 535 DOF.
"""
from __future__ import annotations

from itertools import permutations
from typing import Dict, Tuple

import numpy as np

BOX = (80.0, 10.0, 120.0)
# mid-side node k (4..9) of a C3D10 sits between corners _T10_EDGES[k-4]
_T10_EDGES = [(0, 1), (1, 2), (2, 0), (0, 3), (3, 1), (2, 3)]


def _kuhn_local():
    """6 tets of the unit cube as local corner ids (bit0=x, bit1=y, bit2=z), positively oriented."""
    corner = lambda v: int(v[0]) + 2 * int(v[1]) + 4 * int(v[2])
    dN = np.array([[0., 0., 1.], [1., 0., 0.], [-1., -1., -1.], [0., 1., 0.]])
    tets = []
    for perm in permutations(range(3)):
        v = np.zeros(3)
        verts = [v.copy()]
        for ax in perm:
            v[ax] += 1
            verts.append(v.copy())
        verts = np.array(verts)
        order = [0, 1, 2, 3]
        if np.linalg.det(verts[order].T @ dN) < 0:
            order = [1, 0, 2, 3]
        assert np.linalg.det(verts[order].T @ dN) > 0
        tets.append([corner(verts[i]) for i in order])
    return np.array(tets, dtype=np.int64)


def plate_grid(nx: int, ny: int, nz: int, box: Tuple[float, float, float] = BOX):
    """nodes f64[(nx+1)(ny+1)(nz+1), 3] (x fastest, then y, then z) and tets i32[6*nx*ny*nz, 4]."""
    xs = np.linspace(0.0, box[0], nx + 1)
    ys = np.linspace(0.0, box[1], ny + 1)
    zs = np.linspace(0.0, box[2], nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    nodes = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    sx, sy, sz = 1, nx + 1, (nx + 1) * (ny + 1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    idt = np.int32 if nodes.shape[0] < 2 ** 31 else np.int64
    base = (ix * sx + iy * sy + iz * sz).ravel().astype(idt)           # cell origin node, cells x-fastest
    corner_off = np.array([(c & 1) * sx + ((c >> 1) & 1) * sy + ((c >> 2) & 1) * sz for c in range(8)], dtype=idt)
    local = _kuhn_local()                                              # [6,4] corner ids
    tets = base[:, None, None] + corner_off[local][None, :, :]         # [ncell,6,4]
    return nodes, tets.reshape(-1, 4).astype(np.int32, copy=False)


def to_quadratic(nodes: np.ndarray, tets: np.ndarray):
    """C3D4 -> C3D10: one new node per unique edge, reference mid-side ordering 4..9."""
    t = tets.astype(np.int64)
    pairs = np.stack([np.stack([t[:, a], t[:, b]], axis=1) for a, b in _T10_EDGES], axis=1)   # [ne,6,2]
    key = np.sort(pairs.reshape(-1, 2), axis=1)
    nn = nodes.shape[0]
    code = key[:, 0] * nn + key[:, 1]
    uniq, inv = np.unique(code, return_inverse=True)
    mid = 0.5 * (nodes[uniq // nn] + nodes[uniq % nn])
    t10 = np.concatenate([t, nn + inv.reshape(-1, 6)], axis=1)
    return np.concatenate([nodes, mid], axis=0), t10.astype(np.int32)


def spatial_renumber(nodes: np.ndarray, el: np.ndarray):
    """renumber nodes in (z, y, x) lexicographic order: mid-side nodes of a quadratic mesh then sit next to the
    corner nodes they connect (to_quadratic appends them after all corners), which is what the SpMV's x-gathers
    and the sliced-ELL padding like."""
    order = np.lexsort((nodes[:, 0], nodes[:, 1], nodes[:, 2]))
    new_id = np.empty(order.size, dtype=np.int64)
    new_id[order] = np.arange(order.size)
    return np.ascontiguousarray(nodes[order]), new_id[el].astype(np.int32)


def plate_slab(nx: int, ny: int, nz: int, z0: int, z1: int, box: Tuple[float, float, float] = BOX):
    """cell layers [z0, z1) of the (nx, ny, nz) plate, built without the global mesh: local nodes (the node planes
    z0 .. z1, in global order), local tets (the same Kuhn split and node order as `plate_grid`, so a slab is
    element for element the corresponding range of the global mesh) and l2g (global id of every local node)."""
    assert 0 <= z0 < z1 <= nz
    xs = np.linspace(0.0, box[0], nx + 1)
    ys = np.linspace(0.0, box[1], ny + 1)
    zs = np.linspace(0.0, box[2], nz + 1)[z0:z1 + 1]
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    nodes = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    plane = (nx + 1) * (ny + 1)
    sx, sy, sz = 1, nx + 1, plane
    iz, iy, ix = np.meshgrid(np.arange(z1 - z0), np.arange(ny), np.arange(nx), indexing="ij")
    base = (ix * sx + iy * sy + iz * sz).ravel().astype(np.int32)
    corner_off = np.array([(c & 1) * sx + ((c >> 1) & 1) * sy + ((c >> 2) & 1) * sz for c in range(8)], dtype=np.int32)
    tets = base[:, None, None] + corner_off[_kuhn_local()][None, :, :]
    l2g = np.arange(z0 * plane, (z1 + 1) * plane, dtype=np.int64)
    return nodes, tets.reshape(-1, 4).astype(np.int32, copy=False), l2g


def twist_plate(nx: int, ny: int, nz: int, quadratic: bool = False, renumber: bool = False) -> Dict:
    """the twist-plate model on an (nx,ny,nz)-cell grid, in the reader's vocabulary."""
    nodes, el = plate_grid(nx, ny, nz)
    if quadratic:
        nodes, el = to_quadratic(nodes, el)
        if renumber:
            nodes, el = spatial_renumber(nodes, el)
    tol = 1e-9 * BOX[2]
    clamp = np.nonzero(np.abs(nodes[:, 2] - BOX[2]) < tol)[0]
    twist = np.nonzero(np.abs(nodes[:, 2]) < tol)[0]
    etype = "C3D10" if quadratic else "C3D4"
    dirichlet = ([{"node_set": clamp, "dof": d, "val": 0.0, "user": False} for d in range(3)] +
                 [{"node_set": twist, "dof": d, "val": 0.0, "user": True} for d in range(3)])
    return {"nodes": nodes, "elements": el, "etype": etype, "node_sets": {"Set-10": clamp, "fit_right_z": twist},
            "dirichlet_bc_info": dirichlet, "neumann_bc_info": [],
            "elastic": (2.0e11, 0.3), "geometric_nonlinear": True,
            "time_incs": {"ini_inc": 0.05, "max_time": 1.0, "min_inc": 1e-5, "max_inc": 0.05},
            "cells": (nx, ny, nz)}


def twist_plate_bcs(nodes: np.ndarray):
    """the twist model's *Boundary blocks on a (sub-)mesh of the plate: (dirichlet_bc_info, node_sets) with node ids
    local to `nodes` -- a z-slab that touches neither end face gets empty sets."""
    tol = 1e-9 * BOX[2]
    clamp = np.nonzero(np.abs(nodes[:, 2] - BOX[2]) < tol)[0]
    twist = np.nonzero(np.abs(nodes[:, 2]) < tol)[0]
    dirichlet = ([{"node_set": clamp, "dof": d, "val": 0.0, "user": False} for d in range(3)] +
                 [{"node_set": twist, "dof": d, "val": 0.0, "user": True} for d in range(3)])
    return dirichlet, {"Set-10": clamp, "fit_right_z": twist}


def twist_plate_k(k: int, quadratic: bool = False, renumber: bool = False) -> Dict:
    """BASELINE.md family: cells (8k, k, 12k)."""
    return twist_plate(8 * k, k, 12 * k, quadratic, renumber)


def scaling_cells(n_gpus: int) -> Tuple[int, int, int]:
    """cell grids with exactly 995 328 C3D4 elements per rank under a z-slab partition
    (1: 96x12x144 = BASELINE k=12;  8: 192x24x288 = BASELINE k=24)."""
    table = {1: (96, 12, 144), 2: (96, 24, 144), 4: (192, 24, 144), 8: (192, 24, 288)}
    if n_gpus in table:
        return table[n_gpus]
    return (96, 12, 144 * n_gpus)


def beam_quad8(nx: int = 40, ny: int = 4, length: float = 40.0, height: float = 4.0, plane: str = "CPE8",
               tip_disp: float = 20.0) -> Dict:
    """BASELINE configs[1] stand-in (no CPE8 beam deck is shipped): the 40 x 4 beam of
    tests/beam_deflection (E = 2e5, nu = 0.3, nlgeom=YES, left face clamped, right side u_x = 0 and
    u_y = tip_disp) meshed with nx x ny serendipity quadrilaterals (corner order 0-3 counter-clockwise,
    mid-sides 4:(0,1) 5:(1,2) 6:(2,3) 7:(3,0), element_quadratic_quadrilateral.py:7-14)."""
    gx, gy = 2 * nx + 1, 2 * ny + 1
    I, J = np.meshgrid(np.arange(gx), np.arange(gy), indexing="xy")          # [gy, gx]
    used = ~((I % 2 == 1) & (J % 2 == 1))
    ids = -np.ones((gy, gx), dtype=np.int64)
    ids[used] = np.arange(used.sum())
    nodes = np.stack([I[used] * (length / (2 * nx)), J[used] * (height / (2 * ny))], axis=1).astype(np.float64)
    ex, ey = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    i0, j0 = 2 * ex.ravel(), 2 * ey.ravel()
    loc = [(0, 0), (2, 0), (2, 2), (0, 2), (1, 0), (2, 1), (1, 2), (0, 1)]
    el = np.stack([ids[j0 + dj, i0 + di] for di, dj in loc], axis=1).astype(np.int32)
    left = np.nonzero(nodes[:, 0] < 1e-9)[0]
    right = np.nonzero(nodes[:, 0] > length - 1e-9)[0]
    dirichlet = [{"node_set": right, "dof": 0, "val": 0.0, "user": False},
                 {"node_set": left, "dof": 0, "val": 0.0, "user": False},
                 {"node_set": left, "dof": 1, "val": 0.0, "user": False},
                 {"node_set": right, "dof": 1, "val": tip_disp, "user": False}]
    return {"nodes": nodes, "elements": el, "etype": plane, "node_sets": {"left_face": left, "right_side": right},
            "dirichlet_bc_info": dirichlet, "neumann_bc_info": [], "elastic": (2.0e5, 0.3),
            "geometric_nonlinear": True,
            "time_incs": {"ini_inc": 0.25, "max_time": 1.0, "min_inc": 1e-5, "max_inc": 0.25},
            "bc_blocks": [(False, ["right_side, 1, 1"]), (False, ["left_face, 1, 1", "left_face, 2, 2"]),
                          (False, ["right_side, 2, 2, %.17g" % tip_disp])]}


def write_inp(path: str, mesh: Dict, part: str = "Part-1"):
    """write a reader-compatible Abaqus deck (same keyword layout as the shipped decks)."""
    nodes, el = mesh["nodes"], mesh["elements"]
    inst = f"{part}-1"
    blocks = mesh.get("bc_blocks", [(False, ["Set-10, %d, %d" % (d, d) for d in (1, 2, 3)]),
                                    (True, ["fit_right_z, %d, %d" % (d, d) for d in (1, 2, 3)])])
    with open(path, "w") as f:
        f.write("*Heading\n** generated by femcy_amd.meshgen\n*Part, name=%s\n*End Part\n" % part)
        f.write("*Assembly, name=Assembly\n*Instance, name=%s, part=%s\n*Node\n" % (inst, part))
        for i, p in enumerate(nodes):
            f.write(", ".join(["%d" % (i + 1)] + ["%.17g" % v for v in p]) + "\n")
        f.write("*Element, type=%s\n" % mesh["etype"])
        for i, e in enumerate(el):
            f.write(", ".join(str(v) for v in [i + 1] + (e + 1).tolist()) + "\n")
        f.write("*End Instance\n")
        for name, ids in mesh["node_sets"].items():
            f.write("*Nset, nset=%s, instance=%s\n" % (name, inst))
            ids1 = (np.asarray(ids) + 1).tolist()
            for s in range(0, len(ids1), 16):
                f.write(", ".join(str(v) for v in ids1[s:s + 16]) + "\n")
        if "neo_hookean" in mesh:      # (C1, D1): Abaqus writes C10 and the inverse of D1 (reader: D1 = 1 / value)
            c1, d1 = mesh["neo_hookean"]
            f.write("*End Assembly\n*Material, name=Material-1\n*Hyperelastic, neo hooke\n%.17g, %.17g\n" % (c1, 1. / d1))
        else:
            f.write("*End Assembly\n*Material, name=Material-1\n*Elastic\n%.17g, %.17g\n" % mesh["elastic"])
        f.write("*Step, name=Step-1, nlgeom=%s\n*Static\n" % ("YES" if mesh["geometric_nonlinear"] else "NO"))
        t = mesh["time_incs"]
        f.write("%.17g, %.17g, %.17g, %.17g\n" % (t["ini_inc"], t["max_time"], t["min_inc"], t["max_inc"]))
        for user, lines in blocks:
            f.write("*Boundary, user\n" if user else "*Boundary\n")
            f.write("".join(l + "\n" for l in lines))
        f.write("*End Step\n")
