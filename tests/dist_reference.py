"""CPU mirror of the multi-rank PCG of femcy_amd/csrc/kernels_pcg.hip (pcg_solve, multi path): same
collectives in the same places (one packed interface all-reduce with d.Ad appended, one all-gather of
the (r.M.r, max|r|) pair), executed with numpy + torch.distributed so that the partition / owner /
interface logic can be exercised with the gloo backend on CPU (world_size >= 2)."""
import numpy as np
import torch
import torch.distributed as dist


def _allreduce_sum(a: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(a))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def _allgather_pair(s: float, m: float):
    t = torch.tensor([s, m], dtype=torch.float64)
    out = [torch.zeros(2, dtype=torch.float64) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    g = torch.stack(out).numpy()
    return float(g[:, 0].sum()), float(g[:, 1].max())


def iface_sum(part, v: np.ndarray) -> np.ndarray:
    buf = np.zeros(part.niface_global)
    buf[part.iface_global_slot] = v[part.iface_local_dofs]
    buf = _allreduce_sum(buf)
    v = v.copy()
    v[part.iface_local_dofs] = buf[part.iface_global_slot]
    return v


def distributed_pcg(part, K_loc, b_loc, eps=1e-3, maxit=None):
    """K_loc: rank-local sub-assembled CSR matrix; b_loc: consistent (already interface-summed) rhs."""
    own = part.owner.astype(np.float64)
    M = 1.0 / iface_sum(part, K_loc.diagonal())
    x = np.zeros_like(b_loc)
    r = b_loc.copy()
    d = M * r
    rMr, r0 = _allgather_pair(float(np.sum(own * r * M * r)), float(np.abs(r).max()))
    rmax, it = r0, 0
    n_global_bound = maxit if maxit is not None else 10 ** 9
    for i in range(n_global_bound):
        Ad = K_loc @ d
        buf = np.zeros(part.niface_global + 1)
        buf[part.iface_global_slot] = Ad[part.iface_local_dofs]
        buf[-1] = float(d @ Ad)                       # d^T A d = sum_r d_r^T K_r d_r : no owner mask
        buf = _allreduce_sum(buf)
        Ad[part.iface_local_dofs] = buf[part.iface_global_slot]
        alpha = rMr / buf[-1]
        x = x + alpha * d
        r = r - alpha * Ad
        rMr_new, rmax = _allgather_pair(float(np.sum(own * r * M * r)), float(np.abs(r).max()))
        beta = rMr_new / rMr
        d = M * r + beta * d
        rMr = rMr_new
        it = i + 1
        if rmax < eps * r0:
            break
    return x, it, r0, rmax
