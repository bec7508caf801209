"""CPU suite: the N>1 path.  Partition invariants in-process, and a world_size-2 gloo run of the
distributed PCG mirror (tests/dist_reference.py) against the single-rank oracle."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

from femcy_amd import meshgen, partition
from oracle import femcy_oracle as orc
from oracle.elements import elem_def


def small_problem():
    m = meshgen.twist_plate(4, 2, 6)
    ed = elem_def("C3D4")
    mat = orc.Material("lin3d", m["elastic"])
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    return m, ed, mat, cons


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_partition_invariants(nranks):
    m, ed, mat, cons = small_problem()
    nodes, el = m["nodes"], m["elements"]
    parts = partition.build_all_parts(nodes, el, nranks)
    assert sorted(np.concatenate([p.elem_ids for p in parts]).tolist()) == list(range(el.shape[0]))
    sizes = [p.elem_ids.size for p in parts]
    assert max(sizes) - min(sizes) <= 1
    n = nodes.size
    owners = np.zeros(n, dtype=int)
    slot_to_gdof = {}
    for p in parts:
        assert np.array_equal(nodes[p.l2g], p.nodes)
        assert np.array_equal(p.l2g[p.elements], el[p.elem_ids])
        gd = (p.l2g[:, None] * 3 + np.arange(3)).ravel()
        owners[gd] += p.owner
        for ld, sl in zip(p.iface_local_dofs, p.iface_global_slot):
            assert slot_to_gdof.setdefault(int(sl), int(gd[ld])) == int(gd[ld])     # same slot <-> same global DOF
        assert p.niface_global == parts[0].niface_global
    assert (owners == 1).all()                                   # every DOF counted exactly once
    assert len(slot_to_gdof) == parts[0].niface_global
    # sub-assembly identity: sum_r R_r^T K_r R_r = K
    K = orc.assemble_K(orc.Topology(nodes, el, ed), np.zeros(n), mat.C)
    acc = sp.csr_matrix((n, n))
    for p in parts:
        Kl = orc.assemble_K(orc.Topology(p.nodes, p.elements, ed), np.zeros(p.n_local), mat.C).tocoo()
        gd = (p.l2g[:, None] * 3 + np.arange(3)).ravel()
        acc = acc + sp.coo_matrix((Kl.data, (gd[Kl.row], gd[Kl.col])), shape=(n, n)).tocsr()
    assert abs(acc - K).max() < 1e-12 * abs(K).max()
    # slab partition of the structured plate cuts on cell layers: <= 2 neighbours per rank
    for p in parts if 6 % nranks == 0 else []:
        zs = np.unique(np.round(p.nodes[p.iface_local_dofs[::3] // 3][:, 2], 6))
        assert zs.size <= 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from dist_reference import distributed_pcg, iface_sum
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, ed, mat, cons = small_problem()
        nodes, el = m["nodes"], m["elements"]
        part = partition.build_part(nodes, el, world, rank)
        topo = orc.Topology(part.nodes, part.elements, ed)
        # state S1 of the benchmark: prescribed twist at t = 0.05, residual = f_int with Dirichlet rows zeroed
        u = np.zeros(nodes.size)
        for b in m["dirichlet_bc_info"]:
            if b["user"]:
                orc.user_dirichletBC(u, np.asarray(b["node_set"]), 3, b["dof"], nodes, 0.05)
        ul = part.scatter_global(u)
        f_loc = iface_sum(part, orc.internal_force(topo, ul, mat)[0])          # consistent after the interface sum
        K_loc = orc.assemble_K(topo, ul, mat.C)
        gd = (part.l2g[:, None] * 3 + np.arange(3)).ravel()
        is_cons = np.isin(gd, cons)
        lc = np.nonzero(is_cons)[0]
        K_loc = orc._zero_rows_cols_unit_diag(K_loc, lc).tolil()
        for i in lc:                                                           # unit diagonal from the owner only
            K_loc[i, i] = float(part.owner[i])
        K_loc = K_loc.tocsr()
        f_loc[lc] = 0.0
        x, it, r0, rmax = distributed_pcg(part, K_loc, f_loc, eps=1e-9)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x, it=it, r0=r0, rmax=rmax, f=f_loc)
    finally:
        dist.destroy_process_group()


def test_distributed_pcg_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m, ed, mat, cons = small_problem()
    nodes, el = m["nodes"], m["elements"]
    n = nodes.size
    topo = orc.Topology(nodes, el, ed)
    u = np.zeros(n)
    for b in m["dirichlet_bc_info"]:
        if b["user"]:
            orc.user_dirichletBC(u, np.asarray(b["node_set"]), 3, b["dof"], nodes, 0.05)
    f = orc.internal_force(topo, u, mat)[0]
    K = orc._zero_rows_cols_unit_diag(orc.assemble_K(topo, u, mat.C), cons)
    f[cons] = 0.0
    xo, ito, r0o, rmaxo = orc.pcg_reference(K, f, eps=1e-9)
    parts = partition.build_all_parts(nodes, el, world)
    res = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    assert np.abs(partition.gather_owned(parts, [r["f"] for r in res], n) - f).max() < 1e-9 * np.abs(f).max()
    assert all(int(r["it"]) == int(res[0]["it"]) for r in res)
    assert abs(int(res[0]["it"]) - ito) <= max(2, ito // 50)      # summation order differs (sub-assembly)
    assert abs(float(res[0]["r0"]) - r0o) < 1e-9 * r0o
    x = partition.gather_owned(parts, [r["x"] for r in res], n)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-5
    # replicated interface values agree across ranks
    for p, r in zip(parts, res):
        gd = (p.l2g[:, None] * 3 + np.arange(3)).ravel()
        assert np.abs(r["x"] - x[gd]).max() < 1e-9 * np.abs(x).max()


@pytest.mark.parametrize("nranks,axis", [(2, 2), (4, 2), (8, 2), (3, 0)])
def test_neighbour_lists_pair_up(nranks, axis):
    """the send / recv segments of the neighbour exchange: rank r lists q iff q lists r, with the same global DOFs in
    the same order (what makes an ncclSend on one side meet the ncclRecv of the other entry by entry); their union
    is the rank's interface."""
    from femcy_amd import meshgen, partition
    m = meshgen.twist_plate(6, 3, 16)
    parts = partition.build_all_parts(m["nodes"], m["elements"], nranks, axis=axis)
    for p in parts:
        gd = (p.l2g[:, None] * 3 + np.arange(3)).ravel()
        assert list(p.nb_ranks) == sorted(p.nb_ranks) and p.rank not in p.nb_ranks and p.nb_ptr[0] == 0
        union = set()
        for k, q in enumerate(p.nb_ranks):
            mine = gd[p.nb_dofs[p.nb_ptr[k]:p.nb_ptr[k + 1]]]
            assert mine.size and np.all(np.diff(mine) > 0)
            o = parts[q]
            ko = list(o.nb_ranks).index(p.rank)
            theirs = (o.l2g[:, None] * 3 + np.arange(3)).ravel()[o.nb_dofs[o.nb_ptr[ko]:o.nb_ptr[ko + 1]]]
            assert np.array_equal(mine, theirs)
            union.update(p.nb_dofs[p.nb_ptr[k]:p.nb_ptr[k + 1]].tolist())
        assert union == set(p.iface_local_dofs.tolist())
        if axis == 2:
            assert len(p.nb_ranks) <= 2                      # slabs: at most the ranks below and above


def test_local_deck_splits_boundary_conditions():
    """partition.LocalDeck: every loaded facet lands on exactly one rank (the one holding its element), node sets
    keep exactly the nodes a rank holds, shared nodes appear on every rank that holds them."""
    from helpers import deck
    from femcy_amd import partition
    from femcy_amd.body import Body
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck("cook_3d_linearEl_largeDef.inp"))
    el = inp.eSets["C3D4"]
    parts = partition.build_all_parts(inp.nodes, el, 3, axis=0)
    decks = [partition.LocalDeck(inp, p, Body(p.nodes, p.elements, inp.ELE)) for p in parts]
    for k, nb in enumerate(inp.neumann_bc_info):
        seen = []
        for p, d in zip(parts, decks):
            assert d.neumann_bc_info[k]["traction"] == nb["traction"]
            seen += [tuple(sorted(p.l2g[list(f)].tolist())) for f in d.neumann_bc_info[k]["face_set"]]
        assert sorted(seen) == sorted(tuple(sorted(f)) for f in nb["face_set"])          # once each, none lost
    for k, bc in enumerate(inp.dirichlet_bc_info):
        glob = set(np.asarray(bc["node_set"]).tolist())
        back = set()
        for p, d in zip(parts, decks):
            loc = np.asarray(d.dirichlet_bc_info[k]["node_set"])
            assert set(p.l2g[loc].tolist()) == glob & set(p.l2g.tolist())
            back |= set(p.l2g[loc].tolist())
        assert back == glob and d.dirichlet_bc_info[k]["dof"] == bc["dof"]


@pytest.mark.parametrize("nranks", [1, 2, 3, 4])
def test_slab_local_part_equals_global_partition(nranks):
    """bench.py builds each rank's slab (+ interface tables) without the global mesh; it must be the `Part` the
    global partition gives, field by field, and carry the same boundary-condition node sets."""
    nx, ny, nz = 4, 2, 12
    m = meshgen.twist_plate(nx, ny, nz)
    for r in range(nranks):
        a = partition.build_part(m["nodes"], m["elements"], nranks, r)
        b = partition.plate_slab_part(nx, ny, nz, nranks, r)
        for f in ("elem_ids", "l2g", "nodes", "elements", "iface_local_dofs", "iface_global_slot", "owner", "nb_ranks",
                  "nb_ptr", "nb_dofs"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (nranks, r, f)
        assert a.niface_global == b.niface_global and a.dm == b.dm
        bcs, sets = meshgen.twist_plate_bcs(b.nodes)
        for bc_l, bc_g in zip(bcs, m["dirichlet_bc_info"]):
            assert np.array_equal(bc_l["node_set"], a.localize_nodes(bc_g["node_set"]))
            assert (bc_l["dof"], bc_l["user"]) == (bc_g["dof"], bc_g["user"])
