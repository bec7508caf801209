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


def _lines(path):
    with open(path, "r") as fh:
        return fh.read().split("\n")


def _is_comment(line):
    return line[0:2] == "**"


class InpInfo(InpInfoBase):

    def __init__(self, file) -> None:
        self.nodes, self.eSets = self.read_node_element(file)
        self.node_sets, self.ele_sets = self.read_set(file)
        self.face_sets = self.read_face_set(file)
        self.dirichlet_bc_info, self.neumann_bc_info = self.get_boundary_condition(file)
        self.materials = self.read_material(file)
        self.geometric_nonlinear = self.read_geometric_nonlinear(file)
        self.time_incs = self.read_time_inc(file)

    # ------------------------------------------------------------------ nodes and elements
    def read_node_element(self, fileName):
        lines = _lines(fileName)
        labels, coords = [], []
        in_nodes = False
        for line in lines:
            if "*" in line:
                if in_nodes:
                    break
                in_nodes = ("*Node" in line) or ("*NODE" in line) or ("*node" in line)
                continue
            if in_nodes and line.strip():
                rec = [float(t) for t in line.split(",")]
                labels.append(int(rec[0]))
                coords.append(rec[1:])

        tokens: Dict[str, List[str]] = {}
        current = None
        for line in lines:
            if "*" in line:
                current = None
                if ("*ELEMENT" in line) or ("*Element" in line) or ("*element" in line):
                    if ("TYPE=" in line) or ("type=" in line):
                        current = next((t for t in _TYPE_SCAN_ORDER if t in line), None)
                        if current is not None:
                            tokens.setdefault(current, [])
                continue
            if current is not None:
                body = line.rstrip().rstrip(",")
                if body:
                    tokens[current].extend(body.split(","))
        if len(tokens) > 1:
            print("\033[31;1m there are multiple element types in the file: {} \033[0m".format(list(tokens)))

        eSets = {}
        for eType, toks in tokens.items():
            if eType not in _RECORD:
                print("\033[31;1m Error, element type {} is not found! \033[0m".format(eType))
                sys.exit(1)
            width, keep = _RECORD[eType]
            eSets[eType] = np.array([int(t) for t in toks], dtype=np.int64).reshape((-1, width))[:, keep]

        nodes, eSets = self.sequence_order_of_body(dict(zip(labels, coords)), eSets)
        first = list(eSets.keys())[0]
        if first not in ELEMENT_CLASSES:
            raise ValueError("element type {} has no element class in element_zoo".format(first))
        self.ELE = ELEMENT_CLASSES[first]()
        if len(eSets) != 1:
            raise ValueError("\033[31;1m multiple element types have not been supported now \033[0m")
        return nodes, eSets

    def sequence_order_of_body(self, nodes, eSets):
        """node labels -> 0-based positions in file order; connectivity renumbered accordingly."""
        labels = np.fromiter(nodes.keys(), dtype=np.int64, count=len(nodes))
        lut = np.full(int(labels.max()) + 1, -1, dtype=np.int64)
        lut[labels] = np.arange(labels.size)
        for eType in eSets:
            eSets[eType] = lut[eSets[eType]]
        return np.array(list(nodes.values()), dtype=np.float64), eSets

    # --------------------------------------------------------------------------------- sets
    def read_set(self, fileName):
        node_sets, ele_sets = {}, {}
        target, generate = None, False
        for line in _lines(fileName):
            if not line or _is_comment(line):
                continue
            if line[0] == "*":
                fields = line.split(",")
                if fields[0] in ("*Nset", "*Elset") and "instance" in line:
                    store = node_sets if fields[0] == "*Nset" else ele_sets
                    name = fields[1].split("=")[1]
                    store[name] = set()          # a repeated name starts over, as in the reference
                    target = store[name]
                    generate = "generate" in fields[-1]
                else:
                    target = None
                continue
            if target is None:
                continue
            toks = line.split(",")
            try:
                data = [int(t) for t in toks]
            except ValueError:
                data = [int(t) for t in toks[:-1]]
            if generate:
                target.update(np.arange(data[0], data[1] + data[2], data[2]).tolist())
            else:
                target.update(data)
        as_array = lambda s: np.array(sorted(s), dtype=np.int64) - 1
        return {k: as_array(v) for k, v in node_sets.items()}, {k: as_array(v) for k, v in ele_sets.items()}

    def read_face_set(self, fileName):
        if not hasattr(self, "eSets"):
            self.nodes, self.eSets = self.read_node_element(fileName)
        raw: Dict[str, List[tuple]] = {}
        name = None
        for line in _lines(fileName):
            if not line or _is_comment(line):
                continue
            if line[0] == "*":
                fields = line.split(",")
                if fields[0] == "*Surface":
                    name = fields[2].split("=")[1]
                    raw[name] = []
                else:
                    name = None
                continue
            if name is not None:
                fields = line.split(",")
                raw[name].append((fields[0], fields[1]))

        _, ele_sets = self.read_set(fileName)
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
        mode, user = None, False
        for line in _lines(fileName):
            if not line or _is_comment(line):
                continue
            if line[0] == "*":
                if line[0:9] == "*Boundary":
                    mode, user = "D", ("user" in line)
                elif line[0:7] == "*Dsload":
                    mode = "N"
                else:
                    mode = None
                continue
            f = line.split(",")
            if mode == "D":
                dirichlet.append({"node_set": self.node_sets[f[0]], "dof": int(f[1]) - 1,
                                  "val": float(f[3]) if len(f) >= 4 else 0., "user": user})
            elif mode == "N":
                if len(f) <= 3:       # pressure: positive value pushes against the outward normal
                    neumann.append({"face_set": self.face_sets[f[0]], "traction": -float(f[2])})
                else:                 # TRVEC: magnitude + direction
                    neumann.append({"face_set": self.face_sets[f[0]], "traction": float(f[2]),
                                    "direction": np.array([float(t) for t in f[3:6]])})
        return dirichlet, neumann

    # ---------------------------------------------------------------------------- materials
    def read_material(self, fileName):
        raw = {}
        state, mtype = None, None
        for line in _lines(fileName):
            if not line or _is_comment(line):
                continue
            if line[0] == "*" and line[0:9] == "*Material":
                state = "expect_type"
                continue
            if state == "expect_type":
                mtype = line.split("*")[1]
                state = "data"
                continue
            if state == "data":
                if line[0] != "*":
                    raw[mtype] = [float(t) for t in line.split(",")]
                else:
                    state = None
        ele_type = list(self.eSets.keys())[0]
        family = ele_type[0:3]
        materials = {}
        for key, vals in raw.items():
            if family in ("CPS", "CPE"):
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
        for line in _lines(fileName):
            if line[:5] == "*Step":
                return line.split(",")[-1].split("nlgeom=")[-1] != "NO"
        raise ValueError("no *Step keyword in {}".format(fileName))

    def read_time_inc(self, fileName):
        seen = False
        for line in _lines(fileName):
            if line[:7] == "*Static":
                seen = True
                continue
            if seen:
                if _is_comment(line):
                    continue
                ini, tmax, dmin, dmax = [float(t) for t in line.split(",")][:4]
                return {"ini_inc": min(ini, dmax), "max_time": tmax, "min_inc": dmin, "max_inc": dmax}
        raise ValueError("no *Static data line in {}".format(fileName))
