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
