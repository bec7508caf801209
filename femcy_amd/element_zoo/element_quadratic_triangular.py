"""CPS6/CPE6 quadratic triangle, 3-point rule, half-edge facets
(cf. /root/reference/element_zoo/element_quadratic_triangular.py:26-100)."""
import numpy as np
from .element_base import ElementBase

_HALF_EDGES = [(0, 3), (1, 3), (1, 4), (2, 4), (2, 5), (0, 5)]
_MID = {3: [0.5, 0.5], 4: [0., 0.5], 5: [0.5, 0.]}
_CORNER = {0: [1., 0.], 1: [0., 1.], 2: [0., 0.]}
_EDGE_NORMAL = {3: [1., 1.], 4: [-1., 0.], 5: [0., -1.]}


def _area(c):
    return np.array([c[0], c[1], 1. - c[0] - c[1]])


class Element_quadratic_triangular(ElementBase):
    dm, npe = 2, 6
    _gauss_points = [[2. / 3., 1. / 6.], [1. / 6., 2. / 3.], [1. / 6., 1. / 6.]]
    _gauss_weights = [1. / 6.] * 3
    # each straight edge is two half-edges (corner, mid); points = [mid node, corner node]
    facet_natural_coos = {e: [_MID[e[1]], _CORNER[e[0]]] for e in _HALF_EDGES}
    facet_point_weights = {e: [0.5, 0.5] for e in _HALF_EDGES}
    facet_natural_normals = {e: [_EDGE_NORMAL[e[1]]] * 2 for e in _HALF_EDGES}
    inp_surface_num = [((0, 3), (3, 1)), ((1, 4), (4, 2)), ((2, 5), (5, 0))]
    _tri_split = [(0, 3, 5), (3, 1, 4), (5, 4, 2), (3, 4, 5)]
    # area coordinates of the six nodes w.r.t. the Gauss-point triangle
    _extrap_matrix = (np.array([[5, -1, -1], [-1, 5, -1], [-1, -1, 5],
                                [2, 2, -1], [-1, 2, 2], [2, -1, 2]]) / 3.)

    def shapeFunc_pyscope(self, natCoo):
        L = _area(natCoo)
        return np.concatenate([L * (2. * L - 1.), 4. * L * np.roll(L, -1)])

    def dshape_dnat_pyscope(self, natCoo):
        L0, L1, L2 = _area(natCoo)
        return np.array([[4. * L0 - 1., 0.],
                         [0., 4. * L1 - 1.],
                         [1. - 4. * L2, 1. - 4. * L2],
                         [4. * L1, 4. * L0],
                         [-4. * L1, 4. * (L2 - L1)],
                         [4. * (L2 - L0), -4. * L0]])
