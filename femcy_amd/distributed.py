"""one process per GPU: `python -m torch.distributed.run --nproc-per-node N -m femcy_amd.main deck.inp`.

Every rank reads the same deck, keeps one element partition (z-slabs along the longest axis of the mesh,
`femcy_amd.partition`), joins an RCCL communicator through the C ABI (`femcy_comm_init`; the 128-byte unique id is
broadcast with torch.distributed, which is used for nothing else) and runs the reference's driver unchanged: all
scalars the control flow looks at are collective, so every rank takes the same decisions.  The reference itself is
single-device; this is new work on top of its path."""
import os

import numpy as np

from . import backend as be
from . import partition
from .body import Body


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def wanted() -> bool:
    return world()[0] > 1 or os.environ.get("FEMCY_FORCE_PARTITION") == "1"


def init_process_group():
    """-> (rank, nranks, local_rank, comm_uid); torch.distributed (backend nccl = RCCL) carries only the id."""
    import torch
    import torch.distributed as dist
    nranks, rank, local = world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=nranks, device_id=torch.device("cuda", local))
    box = [be.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return rank, nranks, local, box[0]


def partitioned_system(inp, material, verbose=True, tangent="reference"):
    """-> (system, local deck, part): this rank's share of the deck, ready for `system.solve(local_deck)`."""
    from .stiffnessMtrx import System_of_equations
    rank, nranks, local, uid = init_process_group()
    el = list(inp.eSets.values())[0]
    axis = int(np.argmax(np.ptp(inp.nodes, axis=0)))
    part = partition.build_part(inp.nodes, el, nranks, rank, axis=axis)
    body = Body(part.nodes, part.elements, inp.ELE)
    def gather_blobs(blob):
        import torch.distributed as dist
        blobs = [None] * nranks
        dist.all_gather_object(blobs, blob)
        return blobs

    system = System_of_equations(body, material, inp.geometric_nonlinear, device=local, verbose=verbose, part=part,
                                 comm_uid=uid, tangent=tangent,
                                 exchange=os.environ.get("FEMCY_EXCHANGE", "allreduce"),   # or "neighbour" / "auto"
                                 gather_blobs=gather_blobs)
    return system, partition.LocalDeck(inp, part, body), part


def gather_dof(part, dof_local: np.ndarray, n_global: int):
    """the global displacement vector on rank 0 (None elsewhere): every DOF from its owner."""
    import torch.distributed as dist
    gd = (part.l2g[:, None] * part.dm + np.arange(part.dm)[None, :]).ravel()
    sel = part.owner.astype(bool)
    pieces = [None] * part.nranks if part.rank == 0 else None
    dist.gather_object((gd[sel], np.asarray(dof_local)[sel]), pieces, dst=0)
    if part.rank != 0:
        return None
    out = np.zeros(n_global)
    for idx, vals in pieces:
        out[idx] = vals
    return out
