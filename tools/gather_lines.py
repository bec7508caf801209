"""CPU (numpy): how many 128-byte cache lines one wavefront gather of the SpMV touches, by where the vector lives
(the caller's node order / storage order) and by the internal row order the sigma-windows run over (the caller's
numbering, lexicographic coordinate orders, Morton, reverse Cuthill-McKee).  This is the estimate behind
FEMCY_OPT_PCG_STORAGE_ORDER and FEMCY_OPT_NODE_ORDER; femcy_build_pattern measures the same
quantity on the real pattern (femcy_get_node_order).   usage: python tools/gather_lines.py [c3d10|c3d4]"""
import os
import sys

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import meshgen


def adjacency(nn, el):
    npe = el.shape[1]
    rows = np.repeat(el, npe, axis=1).ravel()
    cols = np.tile(el, (1, npe)).ravel()
    A = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(nn, nn))
    A.sum_duplicates()
    A.sort_indices()
    return A


def lines_for(A, order, sigma=4096, label="", nsamp=200):
    """order: rank -> node; positions = rows sorted by length (stable) inside windows of `sigma` ranks"""
    nn = A.shape[0]
    rowlen = np.diff(A.indptr)
    pos = np.empty(nn, dtype=np.int64)
    node_of = np.empty(nn, dtype=np.int64)
    for a0 in range(0, nn, sigma):
        a1 = min(nn, a0 + sigma)
        w = order[a0:a1]
        idx = w[np.argsort(-rowlen[w], kind="stable")]
        node_of[a0:a1] = idx
        pos[idx] = np.arange(a0, a1)
    nsl = nn // 64
    stored = sum(rowlen[node_of[s * 64:(s + 1) * 64]].max() * 64 for s in range(nsl))
    rng = np.random.default_rng(0)
    tot = {"node": 0, "pos": 0}
    n = 0
    for s in rng.choice(nsl - 1, min(nsamp, nsl - 1), replace=False):
        lanes = node_of[s * 64:(s + 1) * 64]
        L = rowlen[lanes].max()
        cols = np.empty((L, 64), dtype=np.int64)
        for i, a in enumerate(lanes):
            r = A.indices[A.indptr[a]:A.indptr[a + 1]]
            others = r[r != a]
            cols[:, i] = np.concatenate([[a], others, np.full(L - 1 - others.size, a)])     # diagonal first, padding = own node
        for j in range(L):
            for key, c in (("node", cols[j]), ("pos", pos[cols[j]])):
                tot[key] += np.unique(np.concatenate([(c * 24) // 128, (c * 24 + 23) // 128])).size
            n += 1
    print(f"{label:34s} padding {stored / rowlen[:nsl * 64].sum() - 1:6.3f}   lines per gather: vectors in node order "
          f"{tot['node'] / n:5.1f}, in storage order {tot['pos'] / n:5.1f}   (minimum 12)")


def morton(nodes):
    lo = nodes.min(0)
    q = ((nodes - lo) / np.ptp(nodes, axis=0).max() * 1023).astype(np.int64)

    def spread(v):
        v = v & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        return (v | (v << 2)) & 0x9249249
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3d10"
    m = meshgen.twist_plate(48, 6, 72, quadratic=True) if wl == "c3d10" else meshgen.twist_plate(96, 12, 144)
    nodes, el = m["nodes"], m["elements"]
    nn = nodes.shape[0]
    A = adjacency(nn, el)
    lines_for(A, np.arange(nn), label=f"{wl} caller's numbering")
    for name, keys in (("z,y,x (x fastest)", (0, 1, 2)), ("z,x,y", (1, 0, 2)), ("y,z,x", (0, 2, 1))):
        lines_for(A, np.lexsort(tuple(nodes[:, k] for k in keys)), label=f"{wl} lexicographic {name}")
    lines_for(A, morton(nodes), label=f"{wl} Morton")
    lines_for(A, np.asarray(reverse_cuthill_mckee(A.astype(np.int32), symmetric_mode=True)), label=f"{wl} reverse Cuthill-McKee")
