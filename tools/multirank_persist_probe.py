"""persistent PCG across ranks, measured on ONE GPU: the 1 M-element plate as 2 z-slabs = 2 contexts of this process
with 128 workgroups each (both kernels co-resident), mailboxes exchanged by pointer, against (a) one slab alone on 128
workgroups (the same kernel shape without a peer), (b) the same two ranks on the three-launch + collective loop.
usage: python tools/multirank_persist_probe.py [iters=300] -> profiles/r03_multirank_persist_probe.txt"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen, partition
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic

nit = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nranks, wgs = 2, 128
nx, ny, nz = 96, 12, 144
mat = LinearIsotropic(2.0e11, 0.3)
parts = [partition.plate_slab_part(nx, ny, nz, nranks, r) for r in range(nranks)]
uid = be.Context.comm_local_id()
blobs = [None] * nranks
gate = threading.Barrier(nranks)
res = [None] * nranks


def setup(p, comm):
    c = be.Context(0)
    c.set_option(107, wgs)
    c.set_mesh(p.nodes, p.elements)
    c.set_element(Element_linear_tetrahedral())
    c.set_material(mat)
    c.build_pattern()
    if comm:
        c.comm_init(p.rank, p.nranks, uid, p.iface_local_dofs, p.iface_global_slot, p.niface_global, p.owner)
        c.comm_set_neighbours(p)
    return c


def rhs(c, p):
    bcs, _ = meshgen.twist_plate_bcs(p.nodes)
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in bcs if len(b["node_set"])]))
    c.assemble_K(-1)
    gd = (p.l2g[:, None] * 3 + np.arange(3)[None, :]).ravel()
    c.upload(be.VEC_RESIDUAL, np.sin(gd * 0.11) * 1e3)
    c.dirichlet_newton(cons, be.VEC_RESIDUAL)


def timed(c, label, out):
    ts = []
    for rep in range(4):
        t = time.perf_counter()
        it, r0, rmax = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
        ts.append((time.perf_counter() - t) / nit * 1e6)
    tm = c.timing()
    out.append(f"{label}: " + " ".join(f"{t:6.2f}" for t in ts) + f" us/it  rmax {rmax:.6e}  paths 3k/persist "
               f"{tm['solves_three']}/{tm['solves_persist']} timeouts {tm['barrier_timeouts']}")


def rank_main(r):
    out = []
    c = setup(parts[r], True)
    blobs[r] = c.comm_mailbox_export()
    gate.wait()
    c.comm_mailbox_import(blobs)
    agreed = c.comm_persist_agree()
    rhs(c, parts[r])
    out.append(f"rank {r}: {c.ne} elements, {c.n} DOF, agreed {agreed}")
    timed(c, f"rank {r} persistent + mailboxes ({wgs} workgroups per rank)", out)
    c.set_option(be.OPT_PCG_PERSIST_MULTI, 0)
    for ex, nm in ((0, "all-reduce"), (1, "neighbour send/recv")):
        c.set_option(be.OPT_EXCHANGE, ex)
        timed(c, f"rank {r} three launches + in-process collectives ({nm})", out)
    c.close()
    res[r] = out


ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
for t in ths:
    t.start()
for t in ths:
    t.join()
for out in res:
    print("\n".join(out))
# one slab alone, same kernel shape, no communicator
out = []
c = setup(parts[0], False)
rhs(c, parts[0])
timed(c, f"slab 0 alone, persistent, {wgs} workgroups, no communicator", out)
c.set_option(107, 0)
timed(c, "slab 0 alone, persistent, 256 workgroups", out)
c.close()
print("\n".join(out))
