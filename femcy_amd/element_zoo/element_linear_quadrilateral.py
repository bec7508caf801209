"""CPS4/CPE4 bilinear quadrilateral, 2x2 Gauss rule
(cf. /root/reference/element_zoo/element_linear_quadrilateral.py:20-85)."""
import numpy as np
from .element_base import ElementBase

_G = 1. / 3. ** 0.5
_CORNERS = np.array([[-1., -1.], [1., -1.], [1., 1.], [-1., 1.]])


def bilinear(c):
    return (1. + _CORNERS[:, 0] * c[0]) * (1. + _CORNERS[:, 1] * c[1]) / 4.


class Element_linear_quadrilateral(ElementBase):
    dm, npe = 2, 4
    _gauss_points = (_CORNERS * _G).tolist()
    _gauss_weights = [1.] * 4
    facet_natural_coos = {(0, 1): [[-1., -1.], [1., -1.]], (1, 2): [[1., -1.], [1., 1.]],
                          (2, 3): [[1., 1.], [-1., 1.]], (0, 3): [[-1., 1.], [-1., -1.]]}
    facet_point_weights = {k: [0.5, 0.5] for k in facet_natural_coos}
    facet_natural_normals = {(0, 1): [[0., -1.]] * 2, (1, 2): [[1., 0.]] * 2,
                             (2, 3): [[0., 1.]] * 2, (0, 3): [[-1., 0.]] * 2}
    inp_surface_num = [((0, 1),), ((1, 2),), ((2, 3),), ((0, 3),)]
    _tri_split = [(0, 1, 2), (0, 2, 3)]
    _extrap_points = (_CORNERS * 3. ** 0.5).tolist()

    def shapeFunc_pyscope(self, natCoo):
        return bilinear(natCoo)

    def dshape_dnat_pyscope(self, natCoo):
        sx, sy = _CORNERS[:, 0], _CORNERS[:, 1]
        return np.stack([sx * (1. + sy * natCoo[1]), sy * (1. + sx * natCoo[0])], axis=1) / 4.
