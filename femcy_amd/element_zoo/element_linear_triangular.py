"""CPS3/CPE3 constant-strain triangle, N = [xi, eta, 1-xi-eta]
(cf. /root/reference/element_zoo/element_linear_triangular.py:24-73)."""
import numpy as np
from .element_base import ElementBase

_R = 2 ** 0.5 / 2.


class Element_linear_triangular(ElementBase):
    dm, npe = 2, 3
    _gauss_points = [[1. / 3., 1. / 3.]]
    _gauss_weights = [0.5]
    facet_natural_coos = {(0, 1): [[0.5, 0.5]], (1, 2): [[0., 0.5]], (0, 2): [[0.5, 0.]]}
    facet_point_weights = {(0, 1): [1.], (1, 2): [1.], (0, 2): [1.]}
    facet_natural_normals = {(0, 1): [[_R, _R]], (1, 2): [[-1., 0.]], (0, 2): [[0., -1.]]}
    inp_surface_num = [((0, 1),), ((1, 2),), ((2, 0),)]
    _tri_split = [(0, 1, 2)]
    _extrap_matrix = np.ones((3, 1))

    def shapeFunc_pyscope(self, natCoo):
        xi, eta = natCoo[0], natCoo[1]
        return np.array([xi, eta, 1. - xi - eta])

    def dshape_dnat_pyscope(self, natCoo):
        return np.array([[1., 0.], [0., 1.], [-1., -1.]])
