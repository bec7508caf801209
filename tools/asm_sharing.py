"""CPU (numpy): how much of rows4's record traffic an element-major loop over a GROUP of matrix rows could share.

k_assemble_rows4 stages an element's record (960 B of gradients + 32 B of weights for C3D10) once per (row, incident
element) pair and computes the ten blocks of that row -- over an element's ten rows that is each of its 100 blocks
exactly once, so the arithmetic has no redundancy to remove; what an element-major order can save is the staging:
one record load per (GROUP of rows, element) instead of one per (row, element).  This script counts both for groups
the kernel could own -- the two rows of a wave's pair, the eight rows a workgroup finishes together, a wave's sixteen
rows, the whole 64-row slice -- together with the LDS the group's accumulators would need (72 B per stored block).
usage: python tools/asm_sharing.py   (the bench's C3D10 plate, row order as femcy_build_pattern picks it)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from femcy_amd import meshgen
from gather_lines import adjacency


def main():
    nx, ny, nz = (int(v) for v in os.environ.get("CELLS", "48,6,72").split(","))
    m = meshgen.twist_plate(nx, ny, nz, quadratic=True)
    nodes, el = m["nodes"], m["elements"]
    nn = nodes.shape[0]
    A = adjacency(nn, el)
    rowlen = np.diff(A.indptr)
    for label, order in (("caller's numbering", np.arange(nn)),
                         ("coordinate order 1 (the default pick)", np.lexsort((nodes[:, 0], nodes[:, 1], nodes[:, 2])))):
        node_of = np.empty(nn, dtype=np.int64)
        for a0 in range(0, nn, 4096):
            w = order[a0:a0 + 4096]
            node_of[a0:a0 + w.size] = w[np.argsort(-rowlen[w], kind="stable")]
        pos = np.empty(nn, dtype=np.int64)
        pos[node_of] = np.arange(nn)
        pe = pos[el]                                         # [ne, 10] storage positions of an element's rows
        pairs = el.size
        print(f"{label}: {nn} rows, {el.shape[0]} elements, {pairs} (row, element) pairs")
        for gname, key in (("pair of a wave (2 adjacent rows)", lambda p: p // 2),
                           ("8 adjacent rows (one write-out group)", lambda p: p // 8),
                           ("a wave's 16 rows of the slice", lambda p: (p // 64) * 4 + (p % 8) // 2),
                           ("whole slice (64 rows)", lambda p: p // 64)):
            g = key(pe)
            g.sort(axis=1)
            distinct = 1 + (np.diff(g, axis=1) != 0).sum(axis=1)          # groups an element's rows fall into
            loads = int(distinct.sum())
            # accumulator bytes of the largest group: stored blocks of its rows
            gid = key(np.arange(nn))
            blocks = np.bincount(gid, weights=rowlen[node_of].astype(np.float64))
            print(f"   {gname:40s} record loads {loads:9d} = pairs / {pairs / loads:4.2f};  accumulators: "
                  f"max {blocks.max() * 72 / 1024:6.1f} KB, mean {blocks.mean() * 72 / 1024:6.1f} KB per group")


if __name__ == "__main__":
    main()
