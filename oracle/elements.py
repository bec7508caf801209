"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `femcy_amd`.

Self-contained numpy restatement of the per-element-type constants of the reference's
element_zoo (all paths relative to /root/reference):

  CPS3/CPE3  element_zoo/element_linear_triangular.py:24-58 (tables), :62-73 (N, dN)
  CPS4/CPE4  element_zoo/element_linear_quadrilateral.py:20-63, :67-85
  CPS6/CPE6  element_zoo/element_quadratic_triangular.py:26-72, :76-100
  CPS8/CPE8  element_zoo/element_quadratic_quadrilateral.py:21-62, :66-108
  C3D4       element_zoo/element_linear_tetrahedral.py:27-64, :68-82
  C3D10      element_zoo/element_quadratic_tetrahedral.py:33-83, :87-126

Each type is described by an `ElemDef` of plain numpy data and two callables.  The
reference's facet tables (natural coordinates of the facet integration points, their
weights, the natural-space outward normals and the Abaqus S<k> -> local-node map) are
data and are reproduced value for value, including the reference's quirks (e.g. the
CPS8 half-edges (0,7)/(3,7) list the corner coordinate of the *other* corner,
element_quadratic_quadrilateral.py:40).
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Tuple
import numpy as np

_S3 = 1.0 / 3.0 ** 0.5


@dataclass
class ElemDef:
    name: str
    npe: int
    dm: int
    gauss_points: np.ndarray          # [nGP, dm]
    gauss_weights: np.ndarray         # [nGP]
    N: Callable[[np.ndarray], np.ndarray]     # nat -> [npe]
    dN: Callable[[np.ndarray], np.ndarray]    # nat -> [npe, dm]
    facet_natural_coos: Dict[Tuple[int, ...], List[List[float]]]
    facet_point_weights: Dict[Tuple[int, ...], List[float]]
    facet_natural_normals: Dict[Tuple[int, ...], List[List[float]]]
    inp_surface_num: List[Tuple[Tuple[int, ...], ...]]
    extrap: np.ndarray = field(default=None)  # [npe, nGP] Gauss-point -> node extrapolation

    @property
    def nGP(self):
        return self.gauss_points.shape[0]

    @property
    def integPointNum_eachFacet(self):
        return len(next(iter(self.facet_point_weights.values())))

    def dN_table(self):
        """dN[nGP, npe, dm]: the only thing the geometry kernels need."""
        return np.stack([self.dN(g) for g in self.gauss_points])


# ----------------------------------------------------------------------------- triangles
def _tri3_N(c):
    return np.array([c[0], c[1], 1. - c[0] - c[1]])


def _tri3_dN(c):
    return np.array([[1., 0.], [0., 1.], [-1., -1.]])


def _tri6_N(c):
    L = np.array([c[0], c[1], 1. - c[0] - c[1]])
    return np.array([L[0] * (2. * L[0] - 1.), L[1] * (2. * L[1] - 1.), L[2] * (2. * L[2] - 1.),
                     4. * L[0] * L[1], 4. * L[1] * L[2], 4. * L[2] * L[0]])


def _tri6_dN(c):
    L = np.array([c[0], c[1], 1. - c[0] - c[1]])
    return np.array([[4. * L[0] - 1., 0.],
                     [0., 4. * L[1] - 1.],
                     [1. - 4. * L[2], 1. - 4. * L[2]],
                     [4. * L[1], 4. * L[0]],
                     [-4. * L[1], 4. * (L[2] - L[1])],
                     [4. * (L[2] - L[0]), -4. * L[0]]])


# ------------------------------------------------------------------------- quadrilaterals
def _quad4_N(c):
    x, y = c[0], c[1]
    return np.array([(1. - x) * (1. - y), (1. + x) * (1. - y), (1. + x) * (1. + y), (1. - x) * (1. + y)]) / 4.


def _quad4_dN(c):
    x, y = c[0], c[1]
    return np.array([[-(1. - y), -(1. - x)],
                     [(1. - y), -(1. + x)],
                     [(1. + y), (1. + x)],
                     [-(1. + y), (1. - x)]]) / 4.


def _quad8_N(c):
    x, y = c[0], c[1]
    return np.array([(1. - x) * (1. - y) * (-1. - x - y) / 4.,
                     (1. + x) * (1. - y) * (-1. + x - y) / 4.,
                     (1. + x) * (1. + y) * (-1. + x + y) / 4.,
                     (1. - x) * (1. + y) * (-1. - x + y) / 4.,
                     (1. - x ** 2) * (1. - y) / 2.,
                     (1. - y ** 2) * (1. + x) / 2.,
                     (1. - x ** 2) * (1. + y) / 2.,
                     (1. - y ** 2) * (1. - x) / 2.])


def _quad8_dN(c):
    x, y = c[0], c[1]
    return np.array([[-(1. - y) * (-2. * x - y) / 4., -(1. - x) * (-2. * y - x) / 4.],
                     [(1. - y) * (2. * x - y) / 4., -(1. + x) * (-2. * y + x) / 4.],
                     [(1. + y) * (2. * x + y) / 4., (1. + x) * (2. * y + x) / 4.],
                     [-(1. + y) * (-2. * x + y) / 4., (1. - x) * (2. * y - x) / 4.],
                     [-2. * x * (1. - y) / 2., -(1. - x ** 2) / 2.],
                     [(1. - y ** 2) / 2., -2. * y * (1. + x) / 2.],
                     [-2. * x * (1. + y) / 2., (1. - x ** 2) / 2.],
                     [-(1. - y ** 2) / 2., -2. * y * (1. - x) / 2.]])


# ------------------------------------------------------------------------------ tetrahedra
def _tet_bary(c):
    # element_linear_tetrahedral.py:68-71: N = [zeta, xi, 1-xi-eta-zeta, eta]
    return np.array([c[2], c[0], 1. - c[0] - c[1] - c[2], c[1]])


def _tet4_N(c):
    return _tet_bary(c)


def _tet4_dN(c):
    return np.array([[0., 0., 1.], [1., 0., 0.], [-1., -1., -1.], [0., 1., 0.]])


def _tet10_N(c):
    L = _tet_bary(c)
    return np.array([L[0] * (2. * L[0] - 1.), L[1] * (2. * L[1] - 1.),
                     L[2] * (2. * L[2] - 1.), L[3] * (2. * L[3] - 1.),
                     4. * L[0] * L[1], 4. * L[1] * L[2], 4. * L[2] * L[0],
                     4. * L[0] * L[3], 4. * L[3] * L[1], 4. * L[2] * L[3]])


def _tet10_dN(c):
    L = _tet_bary(c)
    return np.array([[0., 0., 4. * L[0] - 1.],
                     [4. * L[1] - 1., 0., 0.],
                     [1. - 4. * L[2], 1. - 4. * L[2], 1. - 4. * L[2]],
                     [0., 4. * L[3] - 1., 0.],
                     [4. * L[0], 0., 4. * L[1]],
                     [4. * (L[2] - L[1]), -4. * L[1], -4. * L[1]],
                     [-4. * L[0], -4. * L[0], 4. * (L[2] - L[0])],
                     [0., 4. * L[0], 4. * L[3]],
                     [4. * L[3], 4. * L[1], 0.],
                     [-4. * L[3], 4. * (L[2] - L[3]), -4. * L[3]]])


def _const(rows, v):
    return {k: [list(v[k])] * rows for k in v}


def _build():
    E = {}
    # ---- CPS3 / CPE3
    E["tri3"] = ElemDef(
        "tri3", 3, 2, np.array([[1. / 3., 1. / 3.]]), np.array([0.5]), _tri3_N, _tri3_dN,
        {(0, 1): [[0.5, 0.5]], (1, 2): [[0., 0.5]], (0, 2): [[0.5, 0.]]},
        {(0, 1): [1.], (1, 2): [1.], (0, 2): [1.]},
        {(0, 1): [[2 ** 0.5 / 2., 2 ** 0.5 / 2.]], (1, 2): [[-1., 0.]], (0, 2): [[0., -1.]]},
        [((0, 1),), ((1, 2),), ((2, 0),)],
        extrap=np.ones((3, 1)))                       # element_linear_triangular.py:226
    # ---- CPS4 / CPE4
    E["quad4"] = ElemDef(
        "quad4", 4, 2, np.array([[-_S3, -_S3], [_S3, -_S3], [_S3, _S3], [-_S3, _S3]]), np.ones(4),
        _quad4_N, _quad4_dN,
        {(0, 1): [[-1., -1.], [1., -1.]], (1, 2): [[1., -1.], [1., 1.]],
         (2, 3): [[1., 1.], [-1., 1.]], (0, 3): [[-1., 1.], [-1., -1.]]},
        {k: [0.5, 0.5] for k in [(0, 1), (1, 2), (2, 3), (0, 3)]},
        _const(2, {(0, 1): (0., -1.), (1, 2): (1., 0.), (2, 3): (0., 1.), (0, 3): (-1., 0.)}),
        [((0, 1),), ((1, 2),), ((2, 3),), ((0, 3),)])
    t = 3. ** 0.5                                      # element_linear_quadrilateral.py:228-238
    E["quad4"].extrap = np.array([_quad4_N(p) for p in [[-t, -t], [t, -t], [t, t], [-t, t]]])
    # ---- CPS6 / CPE6
    E["tri6"] = ElemDef(
        "tri6", 6, 2, np.array([[2. / 3., 1. / 6.], [1. / 6., 2. / 3.], [1. / 6., 1. / 6.]]),
        np.full(3, 1. / 6.), _tri6_N, _tri6_dN,
        {(0, 3): [[0.5, 0.5], [1., 0.]], (1, 3): [[0.5, 0.5], [0., 1.]],
         (1, 4): [[0., 0.5], [0., 1.]], (2, 4): [[0., 0.5], [0., 0.]],
         (2, 5): [[0.5, 0.], [0., 0.]], (0, 5): [[0.5, 0.], [1., 0.]]},
        {k: [0.5, 0.5] for k in [(0, 3), (1, 3), (1, 4), (2, 4), (2, 5), (0, 5)]},
        _const(2, {(0, 3): (1., 1.), (1, 3): (1., 1.), (1, 4): (-1., 0.), (2, 4): (-1., 0.),
                   (2, 5): (0., -1.), (0, 5): (0., -1.)}),
        [((0, 3), (3, 1)), ((1, 4), (4, 2)), ((2, 5), (5, 0))],
        # element_quadratic_triangular.py:296-303: rows = area coordinates of the 6 nodes
        # w.r.t. the Gauss-point triangle
        extrap=np.array([[5. / 3., -1. / 3., -1. / 3.], [-1. / 3., 5. / 3., -1. / 3.],
                         [-1. / 3., -1. / 3., 5. / 3.], [2. / 3., 2. / 3., -1. / 3.],
                         [-1. / 3., 2. / 3., 2. / 3.], [2. / 3., -1. / 3., 2. / 3.]]))
    # ---- CPS8 / CPE8 (2x2 reduced integration)
    E["quad8"] = ElemDef(
        "quad8", 8, 2, np.array([[-_S3, -_S3], [_S3, -_S3], [_S3, _S3], [-_S3, _S3]]), np.ones(4),
        _quad8_N, _quad8_dN,
        {(0, 4): [[-1., -1.], [0., -1.]], (1, 4): [[1., -1.], [0., -1.]],
         (1, 5): [[1., -1.], [1., 0.]], (2, 5): [[1., 1.], [1., 0.]],
         (2, 6): [[1., 1.], [0., 1.]], (3, 6): [[-1., 1.], [0., 1.]],
         (0, 7): [[-1., 1.], [-1., 0.]], (3, 7): [[-1., -1.], [-1., 0.]]},
        {k: [0.5, 0.5] for k in [(0, 4), (1, 4), (1, 5), (2, 5), (2, 6), (3, 6), (0, 7), (3, 7)]},
        _const(2, {(0, 4): (0., -1.), (1, 4): (0., -1.), (1, 5): (1., 0.), (2, 5): (1., 0.),
                   (2, 6): (0., 1.), (3, 6): (0., 1.), (0, 7): (-1., 0.), (3, 7): (-1., 0.)}),
        [((0, 4), (1, 4)), ((1, 5), (2, 5)), ((2, 6), (3, 6)), ((0, 7), (3, 7))])
    # element_quadratic_quadrilateral.py:250-300: bilinear extrapolation from the 4 GPs
    E["quad8"].extrap = np.array([_quad4_N(p) for p in
                                  [[-t, -t], [t, -t], [t, t], [-t, t], [0., -t], [t, 0.], [0., t], [-t, 0.]]])
    # ---- C3D4
    E["tet4"] = ElemDef(
        "tet4", 4, 3, np.array([[0.25, 0.25, 0.25]]), np.array([1. / 6.]), _tet4_N, _tet4_dN,
        {(1, 2, 3): [[1. / 3., 1. / 3., 0.]], (0, 2, 3): [[0., 1. / 3., 1. / 3.]],
         (0, 1, 3): [[1. / 3., 1. / 3., 1. / 3.]], (0, 1, 2): [[1. / 3., 0., 1. / 3.]]},
        {k: [1.] for k in [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]},
        {(1, 2, 3): [[0., 0., -1.]], (0, 2, 3): [[-1., 0., 0.]],
         (0, 1, 3): [[1., 1., 1.]], (0, 1, 2): [[0., -1., 0.]]},
        [((0, 1, 2),), ((0, 1, 3),), ((1, 2, 3),), ((0, 2, 3),)],
        extrap=np.ones((4, 1)))
    # ---- C3D10
    a, b = 0.585410196624968, 0.138196601125010
    f123, f023, f013, f012 = (1, 2, 3, 5, 8, 9), (0, 2, 3, 6, 7, 9), (0, 1, 3, 4, 7, 8), (0, 1, 2, 4, 5, 6)
    c_ = (1. - a) / (a - b); d_ = b / (a - b); x_ = (0.5 - b) / (a - b)
    E["tet10"] = ElemDef(
        "tet10", 10, 3, np.array([[a, b, b], [b, a, b], [b, b, a], [b, b, b]]), np.full(4, 1. / 24.),
        _tet10_N, _tet10_dN,
        {f123: [[1., 0., 0.], [0., 1., 0.], [0., 0., 0.], [0.5, 0.5, 0.], [0., 0.5, 0.], [0.5, 0., 0.]],
         f023: [[0., 1., 0.], [0., 0., 1.], [0., 0., 0.], [0., 0.5, 0.], [0., 0.5, 0.5], [0., 0., 0.5]],
         f013: [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.], [0.5, 0., 0.5], [0., 0.5, 0.5], [0.5, 0.5, 0.]],
         f012: [[1., 0., 0.], [0., 0., 1.], [0., 0., 0.], [0.5, 0., 0.5], [0.5, 0., 0.], [0., 0., 0.5]]},
        {k: [1. / 12.] * 3 + [1. / 4.] * 3 for k in (f123, f023, f013, f012)},
        _const(6, {f123: (0., 0., -1.), f023: (-1., 0., 0.), f013: (1., 1., 1.), f012: (0., -1., 0.)}),
        [(f012,), (f013,), (f123,), (f023,)],
        # element_quadratic_tetrahedral.py:321-339
        extrap=np.array([[-d_, -d_, 1. + c_, 2. * d_ - c_], [1. + c_, -d_, -d_, 2. * d_ - c_],
                         [-d_, -d_, -d_, 1. + 3. * d_], [-d_, 1. + c_, -d_, 2. * d_ - c_],
                         [x_, -d_, x_, 1. - 2. * x_ + d_], [x_, -d_, -d_, 1. + 2. * d_ - x_],
                         [-d_, -d_, x_, 1. + 2. * d_ - x_], [-d_, x_, x_, 1. - 2. * x_ + d_],
                         [x_, x_, -d_, 1. - 2. * x_ + d_], [-d_, x_, -d_, 1. + 2. * d_ - x_]]))
    return E


_DEFS = _build()
# reader/inp_info.py:118-122 (Abaqus type -> class)
ABAQUS_TO_KIND = {"CPS3": "tri3", "CPE3": "tri3", "CPS4": "quad4", "CPE4": "quad4",
                  "CPS6": "tri6", "CPE6": "tri6", "CPS8": "quad8", "CPE8": "quad8",
                  "C3D4": "tet4", "C3D10": "tet10"}


def elem_def(abaqus_type: str) -> ElemDef:
    return _DEFS[ABAQUS_TO_KIND[abaqus_type]]
