"""Element plugin interface of femcy_amd (the drop-in for /root/reference/element_zoo).

`ElementBase` keeps the reference's abstract surface (element_base.py:9-53: shapeFunc,
dshape_dnat, *_pyscope twins, globalNormal, strainMtrx, getMesh, extrapolate and the data
attributes dm, gaussPoints, gaussWeights, integPointNum_eachFacet, facet_natural_coos,
facet_point_weights, facet_natural_normals, inp_surface_num) but is table-driven: a concrete
element only declares its data and its two shape-function callables; everything else is generic
numpy here.  What the HIP kernels consume is `tables()`: dN[nGP][npe][dm], w[nGP] and the Voigt
pattern -- the derivative table depends only on the Gauss point, never on the element
(SURVEY.md 2b), so it is staged once into LDS/constant memory by the device code.
"""
import abc
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ..fields import HostField

VOIGT_2D = 0   # [xx, yy, xy]            (engineering shear)
VOIGT_3D = 1   # [xx, yy, zz, xy, zx, yz]


class ElementBase(abc.ABC):
    #: filled by subclasses -------------------------------------------------------------
    dm: int
    npe: int
    _gauss_points: Sequence[Sequence[float]]
    _gauss_weights: Sequence[float]
    facet_natural_coos: Dict[Tuple[int, ...], List[List[float]]]
    facet_point_weights: Dict[Tuple[int, ...], List[float]]
    facet_natural_normals: Dict[Tuple[int, ...], List[List[float]]]
    inp_surface_num: List[Tuple[Tuple[int, ...], ...]]
    _tri_split: Sequence[Tuple[int, int, int]]      # visualisation triangles per element / per face
    _extrap_points = None                           # natural coords fed to _extrap_basis
    _extrap_matrix = None                           # or an explicit [npe, nGP] matrix

    def __init__(self):
        self.gaussPoints = HostField(self._gauss_points)
        self.gaussWeights = HostField(self._gauss_weights)
        self.gaussPoints_visualize = self.gaussPoints
        self.integPointNum_eachFacet = len(next(iter(self.facet_point_weights.values())))

    # ---- shape functions: subclasses implement the two *_pyscope methods ---------------
    @abc.abstractmethod
    def shapeFunc_pyscope(self, natCoo) -> np.ndarray: ...

    @abc.abstractmethod
    def dshape_dnat_pyscope(self, natCoo) -> np.ndarray: ...

    def shapeFunc(self, natCoo):
        return self.shapeFunc_pyscope(np.asarray(natCoo, dtype=np.float64))

    def dshape_dnat(self, natCoo):
        return self.dshape_dnat_pyscope(np.asarray(natCoo, dtype=np.float64))

    # ---- what the device kernels need --------------------------------------------------
    def tables(self) -> dict:
        gp = np.asarray(self.gaussPoints, dtype=np.float64)
        dN = np.ascontiguousarray(np.stack([self.dshape_dnat_pyscope(p) for p in gp]), dtype=np.float64)
        assert dN.shape == (gp.shape[0], self.npe, self.dm)
        return {"nGP": gp.shape[0], "npe": self.npe, "dm": self.dm, "dN": dN,
                "w": np.ascontiguousarray(self.gaussWeights, dtype=np.float64),
                "voigt_kind": VOIGT_2D if self.dm == 2 else VOIGT_3D}

    def facet_tables(self) -> dict:
        """the facet dictionaries (facet_natural_coos / facet_point_weights / facet_natural_normals, keyed by the
        sorted local node tuple) as the plain arrays femcy_loadset_create takes; facet type = position of the key."""
        keys = list(self.facet_natural_coos.keys())
        nip = self.integPointNum_eachFacet
        coos = np.array([[self.facet_natural_coos[k][i] for i in range(nip)] for k in keys], dtype=np.float64)
        return {"keys": keys, "nft": len(keys), "nfn": len(keys[0]), "nip": nip,
                "ft_nodes": np.ascontiguousarray(keys, dtype=np.int32),
                "N": np.ascontiguousarray([[self.shapeFunc_pyscope(c) for c in row] for row in coos], dtype=np.float64),
                "dN": np.ascontiguousarray([[self.dshape_dnat_pyscope(c) for c in row] for row in coos], dtype=np.float64),
                "normal": np.ascontiguousarray([[self.facet_natural_normals[k][i] for i in range(nip)] for k in keys],
                                               dtype=np.float64),
                "weight": np.ascontiguousarray([[self.facet_point_weights[k][i] for i in range(nip)] for k in keys],
                                               dtype=np.float64)}

    # ---- generic numpy bodies ----------------------------------------------------------
    def strainMtrx(self, dsdx) -> np.ndarray:
        """B(grad N) with the reference's Voigt ordering, shape (s, npe*dm)."""
        g = np.asarray(dsdx, dtype=np.float64)
        npe, dm = g.shape
        if dm == 2:
            B = np.zeros((3, 2 * npe))
            B[0, 0::2], B[1, 1::2] = g[:, 0], g[:, 1]
            B[2, 0::2], B[2, 1::2] = g[:, 1], g[:, 0]
        else:
            B = np.zeros((6, 3 * npe))
            B[0, 0::3], B[1, 1::3], B[2, 2::3] = g[:, 0], g[:, 1], g[:, 2]
            B[3, 0::3], B[3, 1::3] = g[:, 1], g[:, 0]
            B[4, 0::3], B[4, 2::3] = g[:, 2], g[:, 0]
            B[5, 1::3], B[5, 2::3] = g[:, 2], g[:, 1]
        return B

    def globalNormal(self, nodes: np.ndarray, facet: list, integPointId=0):
        """outward unit normal n_g = n_nat (dx/dxi)^-1 (normalised with +1e-30) and
        (facet size) x (facet point weight) for one facet integration point."""
        key = tuple(sorted(facet))
        nat = np.asarray(self.facet_natural_coos[key][integPointId], dtype=np.float64)
        jac = np.asarray(nodes).T @ self.dshape_dnat_pyscope(nat)
        n = np.asarray(self.facet_natural_normals[key][integPointId]) @ np.linalg.inv(jac)
        n = n / (np.linalg.norm(n) + 1.e-30)
        p = np.asarray(nodes)
        if self.dm == 2:
            size = np.linalg.norm(p[key[0]] - p[key[1]])
        else:
            size = 0.5 * np.linalg.norm(np.cross(p[key[1]] - p[key[0]], p[key[2]] - p[key[0]]))
        return n, size * self.facet_point_weights[key][integPointId]

    def getMesh(self, elements: np.ndarray):
        """triangles for drawing, face -> elements map, and the outer surface (vectorised)."""
        el = np.asarray(elements)
        tris = np.sort(np.concatenate([el[:, list(t)] for t in self._tri_split], axis=0), axis=1)
        owner = np.tile(np.arange(el.shape[0]), len(self._tri_split))
        face2ele: Dict[Tuple[int, ...], set] = {}
        for f, e in zip(map(tuple, tris.tolist()), owner.tolist()):
            face2ele.setdefault(f, set()).add(e)
        mesh = np.array(list(face2ele.keys()))
        surfaces = np.array([f for f, es in face2ele.items() if len(es) == 1])
        return mesh, face2ele, surfaces

    def extrap_matrix(self) -> np.ndarray:
        """[npe, nGP]: Gauss-point values -> patch-wise nodal values."""
        if self._extrap_matrix is not None:
            return np.asarray(self._extrap_matrix, dtype=np.float64)
        return np.array([self._extrap_basis(np.asarray(p, dtype=np.float64)) for p in self._extrap_points])

    def _extrap_basis(self, nat):           # overridden where extrapolation uses another basis
        return self.shapeFunc_pyscope(nat)

    def extrapolate(self, internal_vals, nodal_vals, comp: int = 0):
        """Gauss-point values -> patch-wise nodal values (no averaging across elements).  A device
        Gauss-point field (`backend.GaussField`) is extrapolated by the HIP kernel of its context;
        a host array by one matrix product."""
        if hasattr(internal_vals, "ctx"):
            out = internal_vals.ctx.extrapolate(internal_vals.which, self.extrap_matrix(), comp)
        else:
            out = np.asarray(internal_vals) @ self.extrap_matrix().T
        if hasattr(nodal_vals, "from_numpy"):
            nodal_vals.from_numpy(out)
        elif nodal_vals is not None:
            nodal_vals[...] = out
        return out
