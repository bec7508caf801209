"""shared test plumbing: deck paths and the product-reader -> oracle-input adapter."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DECKS = os.path.join(ROOT, "tests", "golden", "decks")
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle.femcy_oracle import Material, OracleSystem  # noqa: E402


def deck(name):
    return os.path.join(DECKS, name)


def oracle_material(mat) -> Material:
    """femcy_amd.material_zoo object -> oracle Material (kind + the two numbers)."""
    kind = {0: "lin3d", 1: "pstrain", 2: "pstress", 3: "neohooke"}[mat.kind]
    return Material(kind, tuple(float(v) for v in mat.params))


def oracle_system_from_inp(inp, **kw) -> OracleSystem:
    etype = list(inp.eSets.keys())[0]
    mat = list(inp.materials.values())[0]
    return OracleSystem(inp.nodes, inp.eSets[etype], etype, oracle_material(mat), inp.geometric_nonlinear, **kw)


def node_adjacency(elements, nn):
    """CSR node adjacency (diagonal included, ascending columns) from the unique element edges -- the same pattern
    COracle derives through a scipy COO of all ne*npe^2 pairs, at a fraction of the memory (8 M elements: 48 M
    edge codes instead of 127 M pairs)."""
    el = np.asarray(elements, dtype=np.int64)
    npe = el.shape[1]
    codes = []
    for a in range(npe):
        for b in range(a + 1, npe):
            lo, hi = np.minimum(el[:, a], el[:, b]), np.maximum(el[:, a], el[:, b])
            codes.append(lo * nn + hi)
    code = np.unique(np.concatenate(codes))
    lo, hi = code // nn, code % nn
    rows = np.concatenate([lo, hi, np.arange(nn)])
    cols = np.concatenate([hi, lo, np.arange(nn)])
    order = np.lexsort((cols, rows))
    ptr = np.zeros(nn + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=nn), out=ptr[1:])
    return ptr, cols[order]
