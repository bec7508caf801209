"""Reference-PRODUCED vectors (round 6): tests/golden/reference_produced.json holds what the reference's own Python code
returns on the reference's own inputs -- its .inp reader on all 47 decks of its tests/ tree, its six element classes'
tables and *_pyscope shape functions, its four material classes' elastic matrices -- generated in the build container by
tests/golden/make_golden_reference.py (imports /root/reference behind a decorator-only taichi stand-in; nothing of the
reference travels but these numbers).  The product's reader / plug-ins and the oracle's tables are held to them here:
rows a16 and g1 of the scope table are pinned on the reference itself, not on a restatement."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import DECKS, GOLDEN
from femcy_amd import element_zoo as ez, material_zoo as mz
from femcy_amd.reader import InpInfo
from oracle import femcy_oracle as orc
from oracle.elements import elem_def

REF = json.load(open(os.path.join(GOLDEN, "reference_produced.json")))


def _deck_map():
    """reference path (relative to its tests/) -> file name under tests/golden/decks (SOURCES.md)"""
    out = {}
    for line in open(os.path.join(DECKS, "SOURCES.md")):
        cells = [c.strip().strip("`") for c in line.split("|")]
        if len(cells) >= 3 and cells[1].endswith(".inp") and cells[2].startswith("tests/"):
            out[cells[2][len("tests/"):]] = cells[1]
    return out


DECK_OF = _deck_map()


def sha(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=dtype).tobytes()).hexdigest()[:24]


def face_set(fs):
    return sorted([int(v) for v in f] for f in fs)


def test_every_reference_deck_has_its_vector():
    assert len(REF["decks"]) == 47 and set(REF["decks"]) == set(DECK_OF)


@pytest.mark.parametrize("ref_path", sorted(REF["decks"]))
def test_product_reader_returns_what_the_reference_reader_returns(ref_path):
    """reader/inp_info.py:18-26: nodes, eSets, node / element / face sets, the two boundary-condition lists, materials,
    nlgeom, time increments -- bit-equal arrays (sha256 of the bytes), equal sets, equal lists of boundary conditions in
    the reference's order"""
    want = REF["decks"][ref_path]
    inp = InpInfo(os.path.join(DECKS, DECK_OF[ref_path]))
    assert list(inp.nodes.shape) == want["nodes"]["shape"] and sha(inp.nodes, np.float64) == want["nodes"]["sha"]
    assert type(inp.ELE).__name__ == want["ELE"]
    assert set(inp.eSets) == set(want["eSets"])
    for k, v in inp.eSets.items():
        assert list(np.shape(v)) == want["eSets"][k]["shape"] and sha(v, np.int64) == want["eSets"][k]["sha"], k
    for name in ("node_sets", "ele_sets"):
        got = getattr(inp, name)
        assert set(got) == set(want[name]), name
        for k in got:
            # members (the reference lists a set in the iteration order of a CPython set of np.int64: not data)
            assert sorted(int(x) for x in got[k]) == sorted(want[name][k]), (name, k)
    assert {k: face_set(v) for k, v in inp.face_sets.items()} == want["face_sets"]
    assert len(inp.dirichlet_bc_info) == len(want["dirichlet_bc_info"])
    for b, w in zip(inp.dirichlet_bc_info, want["dirichlet_bc_info"]):
        assert sorted(int(x) for x in b["node_set"]) == sorted(w["node_set"]) and int(b["dof"]) == w["dof"]
        assert float(b["val"]) == w["val"] and bool(b["user"]) == w["user"]
    assert len(inp.neumann_bc_info) == len(want["neumann_bc_info"])
    for b, w in zip(inp.neumann_bc_info, want["neumann_bc_info"]):
        assert face_set(b["face_set"]) == w["face_set"] and float(b["traction"]) == w["traction"]
        assert [float(x) for x in np.asarray(b.get("direction", []), dtype=float)] == w["direction"]
    assert set(inp.materials) == set(want["materials"])
    for k, m in inp.materials.items():
        assert type(m).__name__ == want["materials"][k]["class"]
        assert np.array_equal(np.asarray(m.C, dtype=float), np.array(want["materials"][k]["C"])), k
    assert bool(inp.geometric_nonlinear) == want["geometric_nonlinear"]
    assert {k: float(v) for k, v in inp.time_incs.items()} == want["time_incs"]


PRODUCT_ELE = {"CPS3": ez.Element_linear_triangular, "CPS4": ez.Element_linear_quadrilateral,
               "CPS6": ez.Element_quadratic_triangular, "CPS8": ez.Element_quadratic_quadrilateral,
               "C3D4": ez.Element_linear_tetrahedral, "C3D10": ez.Element_quadratic_tetrahedral}


@pytest.mark.parametrize("etype", sorted(REF["elements"]))
def test_element_tables_are_the_reference_classes_tables(etype):
    """element_zoo/*.py: Gauss points and weights, shapeFunc_pyscope / dshape_dnat_pyscope at the Gauss points and at
    three fixed points, the four facet tables -- the oracle's ElemDef AND the product plug-in (whose dN table is what the
    HIP kernels consume), both EXACTLY (the reference's expressions evaluated in the same order)"""
    want = REF["elements"][etype]
    ed, prod = elem_def(etype), PRODUCT_ELE[etype]()
    assert type(prod).__name__ == want["class"] and prod.dm == ed.dm == want["dm"]
    gp, gw = np.array(want["gauss_points"]), np.array(want["gauss_weights"])
    tab = prod.tables()
    assert np.array_equal(ed.gauss_points, gp) and np.array_equal(ed.gauss_weights, gw)
    assert np.array_equal(np.asarray(prod.gaussPoints.to_numpy()), gp) and np.array_equal(tab["w"], gw)
    for c, N, dN in zip(want["points"], want["N"], want["dN"]):
        c = np.array(c)
        assert np.array_equal(ed.N(c), np.array(N)) and np.array_equal(ed.dN(c), np.array(dN))
        assert np.array_equal(prod.shapeFunc_pyscope(c), np.array(N)) and np.array_equal(prod.dshape_dnat_pyscope(c), np.array(dN))
    assert np.array_equal(tab["dN"], np.array(want["dN"][:len(gp)]))
    key = lambda k: ",".join(str(int(v)) for v in k)
    for name in ("facet_natural_coos", "facet_point_weights", "facet_natural_normals"):
        for obj in (ed, prod):
            got = {key(k): np.asarray(v, dtype=float).tolist() for k, v in getattr(obj, name).items()}
            assert got == want[name], (name, type(obj).__name__)
    norm = lambda t: [[[int(v) for v in f] for f in s] for s in t]
    assert norm(ed.inp_surface_num) == norm(prod.inp_surface_num) == want["inp_surface_num"]
    assert ed.integPointNum_eachFacet == want["integPointNum_eachFacet"]
    # ELE.globalNormal on one distorted element (stiffnessMtrx.py:369-411 builds the consistent loads from it)
    X = np.array(want["globalNormal_nodes"])
    assert X.shape == (ed.npe, ed.dm) and len(want["globalNormal"]) >= len(want["facet_natural_coos"])
    for tag, (n_ref, aw_ref) in want["globalNormal"].items():
        fk, ip = tag.split(";")
        facet = [int(v) for v in fk.split(",")]
        n, aw = prod.globalNormal(X, facet, int(ip))
        assert np.abs(np.asarray(n) - np.array(n_ref)).max() < 1e-14 and abs(aw - aw_ref) <= 1e-15 * abs(aw_ref), tag
        n2, aw2 = orc.global_normal(ed, X, facet, int(ip)) if hasattr(orc, "global_normal") else (n, aw)
        assert np.abs(np.asarray(n2) - np.array(n_ref)).max() < 1e-14 and abs(aw2 - aw_ref) <= 1e-15 * abs(aw_ref), tag


@pytest.mark.parametrize("name", sorted(REF["materials"]))
def test_material_matrices_are_the_reference_classes_matrices(name):
    """material_zoo/*.py __init__: the elastic matrix C of each class for given constants -- product plug-in and oracle,
    bit for bit"""
    want = REF["materials"][name]
    cls = getattr(mz, want["class"])
    kind = {"LinearIsotropic": "lin3d", "LinearIsotropicPlaneStrain": "pstrain", "LinearIsotropicPlaneStress": "pstress",
            "NeoHookean": "neohooke"}[want["class"]]
    C = np.array(want["C"])
    assert np.array_equal(np.asarray(cls(*want["params"]).C, dtype=float), C)
    assert np.array_equal(np.asarray(orc.Material(kind, tuple(want["params"])).C, dtype=float), C)


@pytest.mark.parametrize("ref_path", sorted(REF["decks"]))
def test_topology_is_what_the_reference_body_computes(ref_path):
    """Body.get_nodeEles / get_coElement_nodes / get_boundary (body.py:165-234) on every deck: the oracle's Topology (node
    adjacency = the block pattern of K, nodeEles, boundary facets) and the product's Body hold exactly the reference's
    sets; the sizes the device pattern is checked against in test_assemble_K (nnzb, longest row, widest nodeEles row) are
    the reference's.  The reference lists a set in CPython's iteration order of a set of numpy integers; for these decks
    that order is recorded too, and the C restatement's `sparseIJ` can be built in it (test_oracle_c.py)."""
    from femcy_amd.body import Body
    want = REF["decks"][ref_path]["topology"]
    inp = InpInfo(os.path.join(DECKS, DECK_OF[ref_path]))
    el = list(inp.eSets.values())[0]
    topo = orc.Topology(inp.nodes, el, elem_def(list(inp.eSets.keys())[0]))
    cnt_c = np.diff(topo.adj_ptr)
    cnt_e = np.diff(topo.nodeEles_ptr)
    assert sha(cnt_c, np.int64) == want["coElement_counts_sha"] and sha(cnt_e, np.int64) == want["nodeEles_counts_sha"]
    assert (int(cnt_c.sum()), int(cnt_c.max()), int(cnt_e.max())) == (want["nnzb"], want["max_row_blocks"], want["max_node_elems"])
    assert sha(topo.adj_idx, np.int64) == want["coElement_sorted_sha"]            # CSR indices are sorted per row
    ne = np.concatenate([np.sort(topo.nodeEles_idx[topo.nodeEles_ptr[a]:topo.nodeEles_ptr[a + 1]]) for a in range(topo.nn)])
    assert sha(ne, np.int64) == want["nodeEles_sorted_sha"]
    b = topo.boundary()
    assert len(b) == want["boundary_facets"]
    assert sha(np.array(sorted([list(k) + [int(v)] for k, v in b.items()]), dtype=np.int64), np.int64) == want["boundary_sha"]
    body = Body(nodes=inp.nodes, elements=el, ELE=inp.ELE)
    assert sha(np.concatenate([np.asarray(x, dtype=np.int64) for x in body.get_coElement_nodes()]), np.int64) == want["coElement_sorted_sha"]
    assert sha(np.concatenate([np.asarray(x, dtype=np.int64) for x in body.get_nodeEles()]), np.int64) == want["nodeEles_sorted_sha"]
    pb = body.get_boundary()
    assert sha(np.array(sorted([list(k) + [int(v)] for k, v in pb.items()]), dtype=np.int64), np.int64) == want["boundary_sha"]


@pytest.mark.parametrize("ref_path", sorted(REF["decks"]))
def test_set_order_emulation_is_the_reference_column_order(ref_path):
    """tests/test_oracle_c.py builds the as-written C restatement's sparseIJ "in the order the reference's Python sets
    produce" by re-running body.py:165-194's set logic (_reference_adjacency): held here to the order the REFERENCE's own
    get_coElement_nodes produced on every deck -- the README reproduction (93.56 / 93.32 / 84.40 at the reference's stop
    iterate) runs on the reference's real column order, not on a guess at it."""
    from test_oracle_c import _reference_adjacency
    want = REF["decks"][ref_path]["topology"]
    inp = InpInfo(os.path.join(DECKS, DECK_OF[ref_path]))
    el = np.asarray(list(inp.eSets.values())[0])
    ptr, idx = _reference_adjacency(el, inp.nodes.shape[0])
    assert sha(idx, np.int64) == want["coElement_reference_order_sha"] and int(ptr[-1]) == want["nnzb"]


NEUMANN_DECKS = sorted(k for k, v in REF["decks"].items() if "neumann_rhs" in v)


def _reference_rhs(want, n):
    out = np.zeros(n)
    out[np.array(want["idx"], dtype=np.int64)] = np.array(want["val"])
    return out


@pytest.mark.parametrize("ref_path", NEUMANN_DECKS)
def test_consistent_loads_are_what_the_reference_neumannBC_computes(ref_path):
    """System_of_equations.neumannBC (stiffnessMtrx.py:369-411) is a plain Python loop in the reference; its load vector of
    every *Dsload of every deck (35 decks: pressures and TRVEC tractions on 2-node edges, half-edges of the quadratic 2-D
    families, 3- and 6-node triangles) was produced by the reference itself.  The oracle's restatement against it --
    entry for entry, 1e-14 of the largest load (the order of the += differs) -- on the product reader's face sets."""
    assert len(NEUMANN_DECKS) == 35
    wants = REF["decks"][ref_path]["neumann_rhs"]
    inp = InpInfo(os.path.join(DECKS, DECK_OF[ref_path]))
    et = list(inp.eSets.keys())[0]
    topo = orc.Topology(inp.nodes, np.asarray(inp.eSets[et]), elem_def(et))
    assert len(wants) == len(inp.neumann_bc_info)
    for nb, want in zip(inp.neumann_bc_info, wants):
        ref = _reference_rhs(want, topo.n)
        got = orc.neumann_rhs(topo, sorted(nb["face_set"]), nb["traction"], nb.get("direction"))
        assert np.abs(got - ref).max() <= 1e-14 * np.abs(ref).max()
        assert np.allclose(got.reshape(-1, topo.dm).sum(axis=0), want["resultant"], rtol=0, atol=1e-12 * np.abs(ref).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("ref_path", NEUMANN_DECKS)
def test_device_loads_are_what_the_reference_neumannBC_computes(ref_path):
    """the same vectors against femcy_loadset_neumann on the device, through the product driver's own facet -> (element,
    facet type) resolution (row a8 of the scope table on a reference-produced vector)"""
    from femcy_amd.body import Body
    from femcy_amd.stiffnessMtrx import System_of_equations
    wants = REF["decks"][ref_path]["neumann_rhs"]
    inp = InpInfo(os.path.join(DECKS, DECK_OF[ref_path]))
    el = list(inp.eSets.values())[0]
    system = System_of_equations(Body(inp.nodes, el, inp.ELE), list(inp.materials.values())[0], inp.geometric_nonlinear, verbose=False)
    try:
        for nb, want in zip(inp.neumann_bc_info, wants):
            system.neumannBC(nb["face_set"], load_val=nb["traction"], load_dir=nb.get("direction", np.array([])))
            got = system.rhs.to_numpy()
            ref = _reference_rhs(want, got.size)
            assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    finally:
        system.ctx.close()
