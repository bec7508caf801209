"""CPS8/CPE8 serendipity quadrilateral with 2x2 reduced integration and half-edge facets
(cf. /root/reference/element_zoo/element_quadratic_quadrilateral.py:21-108)."""
import numpy as np
from .element_base import ElementBase
from .element_linear_quadrilateral import bilinear

_G = 1. / 3. ** 0.5
_T = 3. ** 0.5


class Element_quadratic_quadrilateral(ElementBase):
    dm, npe = 2, 8
    _gauss_points = [[-_G, -_G], [_G, -_G], [_G, _G], [-_G, _G]]
    _gauss_weights = [1.] * 4
    # NB (0,7)/(3,7) carry the opposite corner's coordinate in the reference table (:40);
    # reproduced as data because consistent loads depend on it.
    facet_natural_coos = {(0, 4): [[-1., -1.], [0., -1.]], (1, 4): [[1., -1.], [0., -1.]],
                          (1, 5): [[1., -1.], [1., 0.]], (2, 5): [[1., 1.], [1., 0.]],
                          (2, 6): [[1., 1.], [0., 1.]], (3, 6): [[-1., 1.], [0., 1.]],
                          (0, 7): [[-1., 1.], [-1., 0.]], (3, 7): [[-1., -1.], [-1., 0.]]}
    facet_point_weights = {k: [0.5, 0.5] for k in facet_natural_coos}
    facet_natural_normals = {k: [{4: [0., -1.], 5: [1., 0.], 6: [0., 1.], 7: [-1., 0.]}[k[1]]] * 2
                             for k in facet_natural_coos}
    inp_surface_num = [((0, 4), (1, 4)), ((1, 5), (2, 5)), ((2, 6), (3, 6)), ((0, 7), (3, 7))]
    _tri_split = [(0, 4, 7), (4, 1, 5), (5, 2, 6), (6, 3, 7), (4, 5, 6), (4, 6, 7)]
    _extrap_points = [[-_T, -_T], [_T, -_T], [_T, _T], [-_T, _T], [0., -_T], [_T, 0.], [0., _T], [-_T, 0.]]

    def _extrap_basis(self, nat):       # Gauss-point values are extrapolated bilinearly (:250-300)
        return bilinear(nat)

    def shapeFunc_pyscope(self, nc):
        x, y = nc[0], nc[1]
        return np.array([(1. - x) * (1. - y) * (-1. - x - y) / 4.,
                         (1. + x) * (1. - y) * (-1. + x - y) / 4.,
                         (1. + x) * (1. + y) * (-1. + x + y) / 4.,
                         (1. - x) * (1. + y) * (-1. - x + y) / 4.,
                         (1. - x * x) * (1. - y) / 2.,
                         (1. - y * y) * (1. + x) / 2.,
                         (1. - x * x) * (1. + y) / 2.,
                         (1. - y * y) * (1. - x) / 2.])

    def dshape_dnat_pyscope(self, nc):
        x, y = nc[0], nc[1]
        return np.array([[(1. - y) * (2. * x + y) / 4., (1. - x) * (2. * y + x) / 4.],
                         [(1. - y) * (2. * x - y) / 4., (1. + x) * (2. * y - x) / 4.],
                         [(1. + y) * (2. * x + y) / 4., (1. + x) * (2. * y + x) / 4.],
                         [(1. + y) * (2. * x - y) / 4., (1. - x) * (2. * y - x) / 4.],
                         [-x * (1. - y), -(1. - x * x) / 2.],
                         [(1. - y * y) / 2., -y * (1. + x)],
                         [-x * (1. + y), (1. - x * x) / 2.],
                         [-(1. - y * y) / 2., -y * (1. - x)]])
