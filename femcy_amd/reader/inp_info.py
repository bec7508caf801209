"""Abaqus/CalculiX `.inp` reader, call-compatible with the reference's `InpInfo`
(/root/reference/reader/inp_info.py:14-368) but Taichi-free and single-pass per section.

Attributes after construction (inp_info.py:18-25): nodes, eSets, ELE, node_sets, ele_sets,
face_sets, dirichlet_bc_info, neumann_bc_info, materials, geometric_nonlinear, time_incs.

Reference behaviours that are kept on purpose:
  * only the first `*Node` block is read; node labels are renumbered 0.. in file order (:353-368);
  * the element type is found by substring test in a fixed list order, so CPS6M -> CPS6 (:66-75);
  * only `*Nset`/`*Elset` keyword lines that contain "instance" are kept; `generate` expands
    start,stop,step inclusively (:142-163);
  * `*Surface` data lines `elset, S<k>` map through ELE.inp_surface_num[k-1]; a face set is a
    set of sorted global-node tuples (:199-212);
  * `*Boundary` data `set, d1[, d2[, val]]`: only d1 is used, val = 4th column or 0; the block is
    "user" when the keyword line contains "user"; every block in the file counts (:218-244);
  * `*Dsload` with <= 3 columns is a pressure: traction = -value along the outward normal; with
    more columns traction = value and direction = columns 3..5 (:246-271);
  * `nlgeom` comes from the last comma field of the first `*Step` line (:319-330);
  * `*Static` data -> ini_inc, max_time, min_inc, max_inc with ini_inc clipped to max_inc (:333-350);
  * 2-D elements accept only `*Elastic`; C3D* accept `*Elastic` and `*Hyperelastic, neo hooke`
    with D1 = 1/value (:294-316).
"""
import sys
from typing import Dict, List

import numpy as np

from ..element_zoo import (Element_linear_triangular, Element_linear_quadrilateral,
                           Element_quadratic_triangular, Element_quadratic_quadrilateral,
                           Element_linear_tetrahedral, Element_quadratic_tetrahedral)
from ..material_zoo import (LinearIsotropic, LinearIsotropicPlaneStrain, LinearIsotropicPlaneStress, NeoHookean)
from .inp_info_base import InpInfoBase

# scan order matters: the first type whose name is a substring of the keyword line wins
_TYPE_SCAN_ORDER = ["C3D8", "C3D20", "C3D4", "C3D10", "B31", "C3D6", "CPS3", "CPE3", "CPE4", "CPS4",
                    "CPE8", "CPS8", "CPS6", "CPE6"]
# (integers per record, slice of node columns kept)
_RECORD = {"C3D8": (9, slice(1, 9)), "C3D20": (21, slice(1, 9)), "C3D4": (5, slice(1, 5)),
           "CPE4": (5, slice(1, 5)), "CPS4": (5, slice(1, 5)), "CPS8": (9, slice(1, 9)),
           "CPE8": (9, slice(1, 9)), "C3D10": (11, slice(1, 11)), "B31": (3, slice(1, 3)),
           "CPS3": (4, slice(1, 4)), "CPE3": (4, slice(1, 4)), "C3D6": (7, slice(1, 7)),
           "CPS6": (7, slice(1, 7)), "CPE6": (7, slice(1, 7))}
ELEMENT_CLASSES = {"CPE3": Element_linear_triangular, "CPS3": Element_linear_triangular,
                   "CPE4": Element_linear_quadrilateral, "CPS4": Element_linear_quadrilateral,
                   "CPS6": Element_quadratic_triangular, "CPE6": Element_quadratic_triangular,
                   "CPS8": Element_quadratic_quadrilateral, "CPE8": Element_quadratic_quadrilateral,
                   "C3D4": Element_linear_tetrahedral, "C3D10": Element_quadratic_tetrahedral}


class _Deck:
    """the file, read once: its lines and the positions of the lines that contain '*' (keyword and comment lines).
    Data blocks are the runs between two such lines, so the section readers below visit keyword lines only and
    hand whole data blocks to numpy -- at 1e6 elements the per-line Python loops of the reference
    (inp_info.py:28-113) were the wall-clock, not the solve."""

    def __init__(self, path):
        with open(path, "r") as fh:
            self.lines = fh.read().split("\n")
        self.star = [i for i, line in enumerate(self.lines) if "*" in line]
        self.star.append(len(self.lines))            # sentinel: end of file closes the last block

    def keywords(self):
        """(keyword line, its data lines) for every non-comment keyword line, in file order.  Comment lines are
        transparent (the data of a keyword continue after a `**` line, as in the reference's line loops); a
        '*' inside a data line ends the block."""
        lines, star = self.lines, self.star
        k, last = 0, len(star) - 1
        while k < last:
            line = lines[star[k]]
            if line[0] == "*" and not _is_comment(line):
                block = lines[star[k] + 1:star[k + 1]]
                j = k + 1
                while j < last and _is_comment(lines[star[j]]):
                    block = block + lines[star[j] + 1:star[j + 1]]
                    j += 1
                yield line, block
                k = j
            else:
                k += 1


_deck_cache = {}


def _deck(path) -> "_Deck":
    import os
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    if key not in _deck_cache:
        _deck_cache.clear()                            # one deck at a time
        _deck_cache[key] = _Deck(path)
    return _deck_cache[key]


def _lines(path):
    return _deck(path).lines


def _numbers(block, dtype):
    """all comma-separated numbers of a data block (trailing commas and blank lines tolerated)."""
    body = ",".join(t for t in (line.rstrip().rstrip(",") for line in block) if t)
    return np.fromstring(body, dtype=dtype, sep=",") if body else np.zeros(0, dtype=dtype)


def _is_comment(line):
    return line[0:2] == "**"


class InpInfo(InpInfoBase):

    def __init__(self, file, allow_2d_hyperelastic: bool = False) -> None:
        """allow_2d_hyperelastic: accept `*Hyperelastic, neo hooke` on CPE elements (plane-strain neo-Hookean, an
        extension of this build).  Off by default: the reference rejects any non-`*Elastic` material on 2-D elements
        (:296-299) and so does this reader."""
        self.allow_2d_hyperelastic = allow_2d_hyperelastic
        self.nodes, self.eSets = self.read_node_element(file)
        self.node_sets, self.ele_sets = self.read_set(file)
        self.face_sets = self.read_face_set(file)
        self.dirichlet_bc_info, self.neumann_bc_info = self.get_boundary_condition(file)
        self.materials = self.read_material(file)
        self.geometric_nonlinear = self.read_geometric_nonlinear(file)
        self.time_incs = self.read_time_inc(file)

    # ------------------------------------------------------------------ nodes and elements
    def read_node_element(self, fileName):
        deck = _deck(fileName)
        lines, star = deck.lines, deck.star
        labels, coords = np.zeros(0, dtype=np.int64), np.zeros((0, 3))
        for k in range(len(star) - 1):                       # the first *Node block only (reference :28-45)
            line = lines[star[k]]
            if ("*Node" in line) or ("*NODE" in line) or ("*node" in line):
                block = [t for t in lines[star[k] + 1:star[k + 1]] if t.strip()]
                if block:
                    width = block[0].count(",") + 1
                    rec = _numbers(block, np.float64).reshape(-1, width)
                    labels, coords = rec[:, 0].astype(np.int64), np.ascontiguousarray(rec[:, 1:])
                break

        tokens: Dict[str, List[np.ndarray]] = {}
        for k in range(len(star) - 1):
            line = lines[star[k]]
            if ("*ELEMENT" in line) or ("*Element" in line) or ("*element" in line):
                if ("TYPE=" in line) or ("type=" in line):
                    current = next((t for t in _TYPE_SCAN_ORDER if t in line), None)
                    if current is not None:
                        tokens.setdefault(current, []).append(_numbers(lines[star[k] + 1:star[k + 1]], np.int64))
        if len(tokens) > 1:
            print("\033[31;1m there are multiple element types in the file: {} \033[0m".format(list(tokens)))

        eSets = {}
        for eType, toks in tokens.items():
            if eType not in _RECORD:
                print("\033[31;1m Error, element type {} is not found! \033[0m".format(eType))
                sys.exit(1)
            width, keep = _RECORD[eType]
            eSets[eType] = np.concatenate(toks).reshape((-1, width))[:, keep]

        nodes, eSets = self.sequence_order_of_body((labels, coords), eSets)
        first = list(eSets.keys())[0]
        if first not in ELEMENT_CLASSES:
            raise ValueError("element type {} has no element class in element_zoo".format(first))
        self.ELE = ELEMENT_CLASSES[first]()
        if len(eSets) != 1:
            raise ValueError("\033[31;1m multiple element types have not been supported now \033[0m")
        return nodes, eSets

    def sequence_order_of_body(self, nodes, eSets):
        """node labels -> 0-based positions in file order; connectivity renumbered accordingly."""
        if isinstance(nodes, dict):                           # the reference's calling convention (:353-368)
            labels = np.fromiter(nodes.keys(), dtype=np.int64, count=len(nodes))
            coords = np.array(list(nodes.values()), dtype=np.float64)
        else:
            labels, coords = nodes
            _, first = np.unique(labels, return_index=True)   # a repeated label keeps its first position and
            if first.size != labels.size:                     # its last coordinates, as a dict would
                keep = np.sort(first)
                last = {int(l): i for i, l in enumerate(labels)}
                coords = coords[[last[int(l)] for l in labels[keep]]]
                labels = labels[keep]
        lut = np.full(int(labels.max()) + 1, -1, dtype=np.int64)
        lut[labels] = np.arange(labels.size)
        for eType in eSets:
            eSets[eType] = lut[eSets[eType]]
        return coords, eSets

    # --------------------------------------------------------------------------------- sets
    def read_set(self, fileName):
        node_sets, ele_sets = {}, {}
        for line, block in _deck(fileName).keywords():
            fields = line.split(",")
            if fields[0] in ("*Nset", "*Elset") and "instance" in line:
                store = node_sets if fields[0] == "*Nset" else ele_sets
                name = fields[1].split("=")[1]
                store[name] = set()                  # a repeated name starts over, as in the reference
                if "generate" in fields[-1]:
                    for data_line in block:
                        if data_line:
                            d = _numbers([data_line], np.int64)
                            store[name].update(np.arange(d[0], d[1] + d[2], d[2]).tolist())
                else:
                    store[name].update(_numbers(block, np.int64).tolist())
        as_array = lambda s: np.array(sorted(s), dtype=np.int64) - 1
        return {k: as_array(v) for k, v in node_sets.items()}, {k: as_array(v) for k, v in ele_sets.items()}

    def read_face_set(self, fileName):
        if not hasattr(self, "eSets"):
            self.nodes, self.eSets = self.read_node_element(fileName)
        raw: Dict[str, List[tuple]] = {}
        for line, block in _deck(fileName).keywords():
            fields = line.split(",")
            if fields[0] == "*Surface":
                name = fields[2].split("=")[1]
                raw[name] = []
                for data_line in block:
                    if data_line:
                        f = data_line.split(",")
                        raw[name].append((f[0], f[1]))

        ele_sets = self.ele_sets if hasattr(self, "ele_sets") else self.read_set(fileName)[1]
        conn = self.eSets[list(self.eSets.keys())[0]]
        face_sets = {}
        for sname, entries in raw.items():
            faces = set()
            for elset, tag in entries:
                k = int(tag.split("S")[1]) - 1
                for local in self.ELE.inp_surface_num[k]:
                    keys = np.sort(conn[ele_sets[elset]][:, list(local)], axis=1)
                    faces.update(map(tuple, keys.tolist()))
            face_sets[sname] = faces
        return face_sets

    # ------------------------------------------------------------------ boundary conditions
    def get_boundary_condition(self, fileName):
        if not hasattr(self, "node_sets"):
            self.node_sets, self.ele_sets = self.read_set(fileName)
        if not hasattr(self, "face_sets"):
            self.face_sets = self.read_face_set(fileName)
        dirichlet, neumann = [], []
        for line, block in _deck(fileName).keywords():
            if line[0:9] == "*Boundary":
                user = "user" in line
                for data_line in block:
                    if data_line:
                        f = data_line.split(",")
                        dirichlet.append({"node_set": self.node_sets[f[0]], "dof": int(f[1]) - 1,
                                          "val": float(f[3]) if len(f) >= 4 else 0., "user": user})
            elif line[0:7] == "*Dsload":
                for data_line in block:
                    if not data_line:
                        continue
                    f = data_line.split(",")
                    if len(f) <= 3:       # pressure: positive value pushes against the outward normal
                        neumann.append({"face_set": self.face_sets[f[0]], "traction": -float(f[2])})
                    else:                 # TRVEC: magnitude + direction
                        neumann.append({"face_set": self.face_sets[f[0]], "traction": float(f[2]),
                                        "direction": np.array([float(t) for t in f[3:6]])})
        return dirichlet, neumann

    # ---------------------------------------------------------------------------- materials
    def read_material(self, fileName):
        raw = {}
        expect_type = False
        for line, block in _deck(fileName).keywords():        # *Material, then the type keyword and its data line
            if line[0:9] == "*Material":
                expect_type = True
            elif expect_type:
                expect_type = False
                data = [t for t in block if t]
                if data:
                    raw[line.split("*")[1]] = [float(t) for t in data[-1].split(",")]
        ele_type = list(self.eSets.keys())[0]
        family = ele_type[0:3]
        materials = {}
        for key, vals in raw.items():
            if family == "CPE" and "neo hooke" in key and getattr(self, "allow_2d_hyperelastic", False):
                from ..material_zoo import NeoHookeanPlaneStrain
                materials[key] = NeoHookeanPlaneStrain(C1=vals[0], D1=1. / vals[1])
            elif family in ("CPS", "CPE"):
                if key != "Elastic":
                    raise ValueError("only support linear elastic material for 2d element now.")
                cls = LinearIsotropicPlaneStress if family == "CPS" else LinearIsotropicPlaneStrain
                materials[key] = cls(modulus=vals[0], poisson_ratio=vals[1])
            elif family == "C3D":
                if key == "Elastic":
                    materials[key] = LinearIsotropic(modulus=vals[0], poisson_ratio=vals[1])
                elif "neo hooke" in key:
                    materials[key] = NeoHookean(C1=vals[0], D1=1. / vals[1])
                else:
                    raise ValueError("material type {} has not been supported now".format(key))
        return materials

    # ---------------------------------------------------------------------- step definition
    def read_geometric_nonlinear(self, fileName) -> bool:
        for line, _ in _deck(fileName).keywords():
            if line[:5] == "*Step":
                return line.split(",")[-1].split("nlgeom=")[-1] != "NO"
        raise ValueError("no *Step keyword in {}".format(fileName))

    def read_time_inc(self, fileName):
        for line, block in _deck(fileName).keywords():
            if line[:7] == "*Static":
                data = [t for t in block if t]
                if data:
                    ini, tmax, dmin, dmax = [float(t) for t in data[0].split(",")][:4]
                    return {"ini_inc": min(ini, dmax), "max_time": tmax, "min_inc": dmin, "max_inc": dmax}
                break
        raise ValueError("no *Static data line in {}".format(fileName))
