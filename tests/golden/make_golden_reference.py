"""BUILD-CONTAINER ONLY: writes tests/golden/reference_produced.json -- outputs of the REFERENCE's own Python code on the
reference's own inputs, produced by importing it from /root/reference (a Python reference may be imported here to generate
golden vectors; it cannot travel, the vectors can).

What of the reference runs without Taichi: everything that is plain Python / numpy -- the .inp reader
(reader/inp_info.py), the element classes' tables and *_pyscope shape functions (element_zoo/*.py), the material
classes' elastic matrices (material_zoo/*.py).  They are imported behind the decorator-only `taichi` stand-in of
check_tables_vs_reference.py (decorators = identity, fields = numpy holders; no Taichi kernel is executed or emulated).

Also `System_of_equations.neumannBC` (stiffnessMtrx.py:369-411), which is a plain Python loop in the reference: its consistent
load vector of every *Dsload.  Recorded per deck of /root/reference/tests (47 decks): node / element arrays as shape + sha256 of their bytes, every node
/ element / face set, the boundary-condition lists, material class + parameters + C, nlgeom flag, time increments.
Recorded per element family: Gauss points, weights, N and dN at the Gauss points and at fixed sample points, the facet
tables.  tests/test_reference_produced.py holds the product reader, the product element / material plug-ins and the
oracle's tables to these numbers.

usage: python tests/golden/make_golden_reference.py
"""
import glob
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import check_tables_vs_reference as chk  # noqa: E402

REF = "/root/reference"
SAMPLES = {2: [[0.2, 0.3], [0.1, 0.15], [0.45, 0.05]], 3: [[0.2, 0.3, 0.1], [0.1, 0.15, 0.25], [0.05, 0.5, 0.2]]}
FAMILIES = [("CPS3", "element_linear_triangular", "Element_linear_triangular"),
            ("CPS4", "element_linear_quadrilateral", "Element_linear_quadrilateral"),
            ("CPS6", "element_quadratic_triangular", "Element_quadratic_triangular"),
            ("CPS8", "element_quadratic_quadrilateral", "Element_quadratic_quadrilateral"),
            ("C3D4", "element_linear_tetrahedral", "Element_linear_tetrahedral"),
            ("C3D10", "element_quadratic_tetrahedral", "Element_quadratic_tetrahedral")]


def sha(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=dtype).tobytes()).hexdigest()[:24]


def facet_key(k):
    return ",".join(str(int(v)) for v in k)


def face_set(fs):
    return sorted([int(v) for v in f] for f in fs)


def node_natural_coordinates(e, npe):
    """natural coordinates of the nodes: where shapeFunc_pyscope is the unit vector e_a (found on a lattice of the
    reference element: nodes of all six families lie on multiples of 1/2)"""
    dm = e.dm
    import itertools
    nat = np.full((npe, dm), np.nan)
    for c in itertools.product(np.arange(-1.0, 1.01, 0.5), repeat=dm):
        N = np.asarray(e.shapeFunc_pyscope(np.array(c)), dtype=float)
        a = int(np.argmax(N))
        if abs(N[a] - 1.0) < 1e-12 and np.abs(np.delete(N, a)).max() < 1e-12 and np.isnan(nat[a, 0]):
            nat[a] = c
    assert not np.isnan(nat).any(), "a node of the reference element was not found on the lattice"
    return nat


def matrix_of(m):
    """C of a reference material object: a ti.Matrix built in __init__ -> the stub keeps the rows in `.a`"""
    c = m.C
    return np.asarray(c.a if hasattr(c, "a") else c, dtype=float)


def main():
    sys.modules["taichi"] = chk._taichi_stub()
    for p in (os.path.join(REF, "element_zoo"), os.path.join(REF, "material_zoo"), os.path.join(REF, "reader"),
              os.path.join(REF, "user_defined"), REF):
        sys.path.insert(0, p)
    import inp_info as ref_reader
    import body as ref_body
    import stiffnessMtrx as ref_sys
    out = {"source": "mo-hanxuan/FEMcy checkout at /root/reference, imported behind a decorator-only taichi stand-in",
           "decks": {}, "elements": {}, "materials": {}}
    for path in sorted(glob.glob(os.path.join(REF, "tests", "**", "*.inp"), recursive=True)):
        r = ref_reader.InpInfo(path)
        d = {"nodes": {"shape": list(r.nodes.shape), "sha": sha(r.nodes, np.float64)},
             "ELE": type(r.ELE).__name__,
             "eSets": {k: {"shape": list(np.shape(v)), "sha": sha(v, np.int64)} for k, v in r.eSets.items()},
             "node_sets": {k: [int(x) for x in v] for k, v in r.node_sets.items()},
             "ele_sets": {k: [int(x) for x in v] for k, v in r.ele_sets.items()},
             "face_sets": {k: face_set(v) for k, v in r.face_sets.items()},
             "dirichlet_bc_info": [{"node_set": [int(x) for x in b["node_set"]], "dof": int(b["dof"]), "val": float(b["val"]),
                                    "user": bool(b["user"])} for b in r.dirichlet_bc_info],
             "neumann_bc_info": [{"face_set": face_set(b["face_set"]), "traction": float(b["traction"]),
                                  "direction": [float(x) for x in np.asarray(b.get("direction", []), dtype=float)]}
                                 for b in r.neumann_bc_info],
             "materials": {k: {"class": type(m).__name__, "C": matrix_of(m).tolist()} for k, m in r.materials.items()},
             "geometric_nonlinear": bool(r.geometric_nonlinear),
             "time_incs": {k: float(v) for k, v in r.time_incs.items()}}
        # Body.get_nodeEles / get_coElement_nodes / get_boundary (body.py:165-234): plain Python on the numpy arrays -- called
        # unbound on a holder of np_nodes / np_elements / ELE (Body.__init__ only adds Taichi fields for the GGUI).  The
        # reference lists every set in CPython's set-iteration order; recorded sorted (membership is the data) and, for the
        # adjacency, also in the reference's own order (the column order of its sparseIJ, stiffnessMtrx.py:78-89)
        el = list(r.eSets.values())[0]
        holder = types.SimpleNamespace(np_nodes=np.asarray(r.nodes), np_elements=np.asarray(el), ELE=r.ELE)
        holder.get_nodeEles = lambda redo=False, h=holder: ref_body.Body.get_nodeEles(h, redo)
        node_eles = ref_body.Body.get_nodeEles(holder)
        co_nodes = ref_body.Body.get_coElement_nodes(holder)
        boundary = ref_body.Body.get_boundary(holder)
        cnt_e = np.array([len(x) for x in node_eles], dtype=np.int64)
        cnt_c = np.array([len(x) for x in co_nodes], dtype=np.int64)
        d["topology"] = {
            "nodeEles_counts_sha": sha(cnt_e, np.int64), "coElement_counts_sha": sha(cnt_c, np.int64),
            "nnzb": int(cnt_c.sum()), "max_row_blocks": int(cnt_c.max()), "max_node_elems": int(cnt_e.max()),
            "nodeEles_sorted_sha": sha(np.concatenate([np.sort(np.asarray(x, dtype=np.int64)) for x in node_eles]), np.int64),
            "coElement_sorted_sha": sha(np.concatenate([np.sort(np.asarray(x, dtype=np.int64)) for x in co_nodes]), np.int64),
            "coElement_reference_order_sha": sha(np.concatenate([np.asarray(x, dtype=np.int64) for x in co_nodes]), np.int64),
            "boundary_facets": len(boundary),
            "boundary_sha": sha(np.array(sorted([list(k) + [int(v)] for k, v in boundary.items()]), dtype=np.int64), np.int64)}
        # System_of_equations.neumannBC (stiffnessMtrx.py:369-411: plain Python over the loaded facets, ELE.globalNormal and
        # shapeFunc_pyscope) called unbound on a holder of body / ELE / rhs / dm: the reference's own consistent load vector
        # of every *Dsload of the deck -- the loaded entries (index, value) and the resultant
        if r.neumann_bc_info:
            holder.boundary = boundary
            holder.get_boundary = lambda redo=False, h=holder: h.boundary
            d["neumann_rhs"] = []
            for nb in r.neumann_bc_info:
                rhs = np.zeros(np.asarray(r.nodes).size)
                sysh = types.SimpleNamespace(body=holder, ELE=r.ELE, rhs=rhs, dm=int(np.asarray(r.nodes).shape[1]))
                ref_sys.System_of_equations.neumannBC(sysh, nb["face_set"], nb["traction"], nb.get("direction", np.array([])))
                nz = np.nonzero(rhs)[0]
                d["neumann_rhs"].append({"idx": [int(i) for i in nz], "val": [float(v) for v in rhs[nz]],
                                         "resultant": [float(v) for v in rhs.reshape(-1, sysh.dm).sum(axis=0)]})
        out["decks"][os.path.relpath(path, os.path.join(REF, "tests"))] = d
    for etype, mod, cls in FAMILIES:
        e = getattr(__import__(mod), cls)()
        gp = e.gaussPoints.to_numpy()
        pts = [list(map(float, g)) for g in gp] + SAMPLES[e.dm]
        out["elements"][etype] = {
            "class": cls, "dm": int(e.dm), "gauss_points": gp.tolist(), "gauss_weights": e.gaussWeights.to_numpy().tolist(),
            "points": pts, "N": [np.asarray(e.shapeFunc_pyscope(np.array(c)), dtype=float).tolist() for c in pts],
            "dN": [np.asarray(e.dshape_dnat_pyscope(np.array(c)), dtype=float).tolist() for c in pts],
            "integPointNum_eachFacet": int(e.integPointNum_eachFacet),
            "facet_natural_coos": {facet_key(k): np.asarray(v, dtype=float).tolist() for k, v in e.facet_natural_coos.items()},
            "facet_point_weights": {facet_key(k): np.asarray(v, dtype=float).tolist() for k, v in e.facet_point_weights.items()},
            "facet_natural_normals": {facet_key(k): np.asarray(v, dtype=float).tolist() for k, v in e.facet_natural_normals.items()},
            "inp_surface_num": [[[int(v) for v in f] for f in s] for s in e.inp_surface_num]}
        # ELE.globalNormal (numpy in the reference): outward normal and size x weight of every facet integration point of
        # ONE distorted element -- its nodes = the natural coordinates of the nodes pushed through a fixed smooth map
        npe = len(out["elements"][etype]["N"][0])
        nat = node_natural_coordinates(e, npe)
        X = np.stack([nat[:, d] * (1.0 + 0.2 * d) + 0.07 * np.sin(1.3 * nat[:, (d + 1) % e.dm] + 0.4 * d) + 0.03 * nat[:, 0] * nat[:, -1]
                      for d in range(e.dm)], axis=1) * 1.7 + 0.3
        gn = {}
        for k in e.facet_natural_coos:
            for ip in range(len(e.facet_natural_coos[k])):
                n, aw = e.globalNormal(X, list(k), ip)
                gn[f"{facet_key(k)};{ip}"] = [np.asarray(n, dtype=float).tolist(), float(aw)]
        out["elements"][etype]["globalNormal_nodes"] = X.tolist()
        out["elements"][etype]["globalNormal"] = gn
    import linear_isotropic, linear_isotropic_plane_strain, linear_isotropic_plane_stress, neo_hookean
    for name, cls, params in (("LinearIsotropic", linear_isotropic.LinearIsotropic, (210000.0, 0.3)),
                              ("LinearIsotropic_soft", linear_isotropic.LinearIsotropic, (3.5, 0.4999)),
                              ("LinearIsotropicPlaneStrain", linear_isotropic_plane_strain.LinearIsotropicPlaneStrain, (210000.0, 0.3)),
                              ("LinearIsotropicPlaneStress", linear_isotropic_plane_stress.LinearIsotropicPlaneStress, (210000.0, 0.3)),
                              ("NeoHookean", neo_hookean.NeoHookean, (80.0, 1.0e-3))):
        m = cls(*params)
        out["materials"][name] = {"class": cls.__name__, "params": list(params), "C": matrix_of(m).tolist()}
    path = os.path.join(HERE, "reference_produced.json")
    json.dump(out, open(path, "w"), indent=0, separators=(",", ":"))
    print(f"{path}: {len(out['decks'])} decks, {len(out['elements'])} element families, {len(out['materials'])} materials, "
          f"{os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
