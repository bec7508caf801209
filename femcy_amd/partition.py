"""Element partition of a mesh over the GPUs of one node (new work: the reference is single-device).

The mesh is cut into `nranks` slabs of (nearly) equal element count along one axis (z, the long
axis of the twist plate: 144 or 288 cell layers, <= 2 neighbours per rank).  Each rank owns its
elements and keeps every node they touch; nodes on a cut are replicated.  The rank-local stiffness
matrix is sub-assembled (interface rows hold partial sums), so per CG iteration the interface
entries of y = K_loc d are summed across ranks through one packed "global interface vector"
all-reduce; dot products count every shared DOF once through the `owner` mask (lowest sharing rank).

`Part` carries exactly what femcy_comm_init() needs:
  iface_local_dofs[k]   local scalar DOF that is entry iface_global_slot[k] of the packed vector
  niface_global         length of the packed vector (all cuts, all ranks)
  owner[i]              1 if this rank counts local DOF i in reductions
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np


@dataclass
class Part:
    rank: int
    nranks: int
    elem_ids: np.ndarray           # global element ids owned by this rank
    l2g: np.ndarray                # local node -> global node (ascending)
    nodes: np.ndarray              # local coordinates
    elements: np.ndarray           # local connectivity (int32)
    iface_local_dofs: np.ndarray   # int32
    iface_global_slot: np.ndarray  # int32
    niface_global: int
    owner: np.ndarray              # uint8[n_local]
    dm: int
    # neighbour exchange (alternative to the packed all-reduce): the ranks this one shares nodes with, ascending, and
    # for neighbour q the local DOFs shared with it in ascending GLOBAL DOF order -- rank q lists the same DOFs in the
    # same order, so the two sides of a send / recv pair line up entry by entry
    nb_ranks: np.ndarray = None    # int32[nnb]
    nb_ptr: np.ndarray = None      # int32[nnb + 1]
    nb_dofs: np.ndarray = None     # int32[nb_ptr[-1]]

    @property
    def n_local(self):
        return self.nodes.shape[0] * self.dm

    def localize_nodes(self, global_node_ids) -> np.ndarray:
        """global node ids -> local ids, dropping nodes this rank does not hold."""
        g = np.asarray(global_node_ids, dtype=np.int64)
        pos = np.searchsorted(self.l2g, g)
        pos = np.clip(pos, 0, self.l2g.size - 1)
        return pos[self.l2g[pos] == g]

    def scatter_global(self, v_global: np.ndarray) -> np.ndarray:
        """restrict a global DOF vector to this rank's nodes."""
        return v_global.reshape(-1, self.dm)[self.l2g].ravel()


def element_ranks(nodes: np.ndarray, elements: np.ndarray, nranks: int, axis: int = 2) -> np.ndarray:
    """rank of every element: equal-count slabs ordered by centroid coordinate along `axis`
    (stable, so structured meshes split exactly on cell layers)."""
    ne = elements.shape[0]
    corners = elements[:, :4] if elements.shape[1] > 4 else elements
    cen = np.ascontiguousarray(nodes[:, axis])[corners].mean(axis=1)      # one coordinate only: 3x less traffic
    order = np.argsort(np.round(cen, 9), kind="stable")
    rank_of = np.empty(ne, dtype=np.int32)
    rank_of[order] = (np.arange(ne, dtype=np.int64) * nranks // ne).astype(np.int32)
    return rank_of


def _sharing(elements: np.ndarray, rank_of: np.ndarray, nn: int, nranks: int):
    """boolean [nn, nranks]: node touched by rank."""
    touched = np.zeros((nn, nranks), dtype=bool)
    touched[elements.ravel(), np.repeat(rank_of, elements.shape[1])] = True
    return touched


def build_part(nodes: np.ndarray, elements: np.ndarray, nranks: int, rank: int, axis: int = 2,
               rank_of: np.ndarray = None) -> Part:
    nn, dm = nodes.shape
    elements = np.asarray(elements)
    if rank_of is None:
        rank_of = element_ranks(nodes, elements, nranks, axis)
    touched = _sharing(elements, rank_of, nn, nranks)
    mult = touched.sum(axis=1)
    iface_nodes = np.nonzero(mult >= 2)[0]                       # global ids, ascending -> slot order
    slot_of_node = np.full(nn, -1, dtype=np.int64)
    slot_of_node[iface_nodes] = np.arange(iface_nodes.size)
    owner_rank = np.argmax(touched, axis=1)                      # lowest rank touching the node

    mine = np.nonzero(rank_of == rank)[0]
    l2g = np.nonzero(touched[:, rank])[0]
    g2l = np.full(nn, -1, dtype=np.int64)
    g2l[l2g] = np.arange(l2g.size)
    loc_el = g2l[elements[mine]].astype(np.int32)
    loc_iface = np.nonzero(slot_of_node[l2g] >= 0)[0]            # local node ids on a cut
    comp = np.arange(dm)
    iface_local_dofs = (loc_iface[:, None] * dm + comp[None, :]).ravel().astype(np.int32)
    iface_global_slot = (slot_of_node[l2g[loc_iface]][:, None] * dm + comp[None, :]).ravel().astype(np.int32)
    owner = np.repeat((owner_rank[l2g] == rank).astype(np.uint8), dm)
    nb_ranks, nb_ptr, nb_dofs = [], [0], []
    mine_if = touched[l2g[loc_iface]]                              # [n_iface_local, nranks]
    for q in range(nranks):
        if q == rank:
            continue
        shared = loc_iface[mine_if[:, q]]                          # local node ids, ascending local = ascending global
        if shared.size:
            nb_ranks.append(q)
            nb_dofs.append((shared[:, None] * dm + comp[None, :]).ravel())
            nb_ptr.append(nb_ptr[-1] + shared.size * dm)
    return Part(rank=rank, nranks=nranks, elem_ids=mine, l2g=l2g, nodes=np.ascontiguousarray(nodes[l2g]),
                elements=loc_el, iface_local_dofs=iface_local_dofs, iface_global_slot=iface_global_slot,
                niface_global=int(iface_nodes.size * dm), owner=owner, dm=dm,
                nb_ranks=np.asarray(nb_ranks, dtype=np.int32), nb_ptr=np.asarray(nb_ptr, dtype=np.int32),
                nb_dofs=(np.concatenate(nb_dofs) if nb_dofs else np.zeros(0)).astype(np.int32))


def plate_slab_part(nx: int, ny: int, nz: int, nranks: int, rank: int) -> Part:
    """the z-slab `Part` of rank `rank` of the structured C3D4 twist plate (`meshgen.plate_grid`), built from the
    rank's own cell layers only -- O(local) host memory and time instead of the O(global) of `build_part`, with
    identical contents (tests/test_distributed_cpu.py compares them field by field).  Needs nz % nranks == 0, which
    is what makes `element_ranks` cut on cell layers."""
    from . import meshgen
    assert nz % nranks == 0, "z-slab partition of the plate needs nz divisible by the number of ranks"
    per = nz // nranks
    z0, z1 = rank * per, (rank + 1) * per
    nodes, el, l2g = meshgen.plate_slab(nx, ny, nz, z0, z1)
    plane = (nx + 1) * (ny + 1)
    dm, comp = 3, np.arange(3)
    ne_per = 6 * nx * ny * per
    elem_ids = np.arange(rank * ne_per, (rank + 1) * ne_per, dtype=np.int64)
    # cut planes: global node plane q * per for q = 1 .. nranks-1, slot = (q-1) * plane + in-plane index
    lo_cut, hi_cut = rank > 0, rank < nranks - 1
    nloc = nodes.shape[0]
    inplane = np.arange(plane, dtype=np.int64)
    loc_iface, slots = [], []
    if lo_cut:
        loc_iface.append(inplane)
        slots.append((rank - 1) * plane + inplane)
    if hi_cut:
        loc_iface.append(nloc - plane + inplane)
        slots.append(rank * plane + inplane)
    loc_iface = np.concatenate(loc_iface) if loc_iface else np.zeros(0, dtype=np.int64)
    slots = np.concatenate(slots) if slots else np.zeros(0, dtype=np.int64)
    iface_local_dofs = (loc_iface[:, None] * dm + comp[None, :]).ravel().astype(np.int32)
    iface_global_slot = (slots[:, None] * dm + comp[None, :]).ravel().astype(np.int32)
    owner = np.ones(nloc * dm, dtype=np.uint8)
    if lo_cut:
        owner[:plane * dm] = 0                                      # the lower neighbour (lower rank) owns the cut
    nb_ranks, nb_ptr, nb_dofs = [], [0], []
    for q, ids in ((rank - 1, inplane if lo_cut else None), (rank + 1, (nloc - plane + inplane) if hi_cut else None)):
        if ids is not None:
            nb_ranks.append(q)
            nb_dofs.append((ids[:, None] * dm + comp[None, :]).ravel())
            nb_ptr.append(nb_ptr[-1] + ids.size * dm)
    return Part(rank=rank, nranks=nranks, elem_ids=elem_ids, l2g=l2g, nodes=nodes, elements=el,
                iface_local_dofs=iface_local_dofs, iface_global_slot=iface_global_slot,
                niface_global=int((nranks - 1) * plane * dm), owner=owner, dm=dm,
                nb_ranks=np.asarray(nb_ranks, dtype=np.int32), nb_ptr=np.asarray(nb_ptr, dtype=np.int32),
                nb_dofs=(np.concatenate(nb_dofs) if nb_dofs else np.zeros(0)).astype(np.int32))


def build_all_parts(nodes, elements, nranks, axis=2) -> List[Part]:
    rank_of = element_ranks(nodes, elements, nranks, axis)
    return [build_part(nodes, elements, nranks, r, axis, rank_of) for r in range(nranks)]


def gather_owned(parts: List[Part], local_vectors: List[np.ndarray], n_global: int) -> np.ndarray:
    """assemble a global vector from per-rank vectors, taking each shared DOF from its owner."""
    out = np.zeros(n_global)
    for p, v in zip(parts, local_vectors):
        gd = (p.l2g[:, None] * p.dm + np.arange(p.dm)[None, :]).ravel()
        sel = p.owner.astype(bool)
        out[gd[sel]] = np.asarray(v)[sel]
    return out


class LocalDeck:
    """what `System_of_equations.solve` reads from an `InpInfo` (dirichlet_bc_info, neumann_bc_info, time_incs),
    restricted to one rank's sub-mesh: node sets become local node ids (nodes the rank does not hold are
    dropped), face sets keep the facets of elements this rank holds, as local sorted node tuples."""

    def __init__(self, inp, part: Part, body):
        self.time_incs = inp.time_incs
        self.geometric_nonlinear = inp.geometric_nonlinear
        self.materials = inp.materials
        self.ELE = inp.ELE
        self.nodes = part.nodes
        self.dirichlet_bc_info = [dict(bc, node_set=part.localize_nodes(np.asarray(bc["node_set"])))
                                  for bc in inp.dirichlet_bc_info]
        g2l = {int(g): i for i, g in enumerate(part.l2g)}
        boundary = body.get_boundary() if inp.neumann_bc_info else {}
        self.neumann_bc_info = []
        for nb in inp.neumann_bc_info:
            faces = set()
            for f in nb["face_set"]:
                if all(int(v) in g2l for v in f):
                    lf = tuple(sorted(g2l[int(v)] for v in f))
                    if lf in boundary:
                        faces.add(lf)
            self.neumann_bc_info.append(dict(nb, face_set=faces))
