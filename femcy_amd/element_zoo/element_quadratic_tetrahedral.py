"""C3D10 quadratic tetrahedron, 4-point rule (w = 1/24), 6-point facet rule
(cf. /root/reference/element_zoo/element_quadratic_tetrahedral.py:33-126)."""
import numpy as np
from .element_base import ElementBase
from .element_linear_tetrahedral import TET_FACES, TET_DBARY, tet_bary

_A, _B = 0.585410196624968, 0.138196601125010
_EDGES = [(0, 1), (1, 2), (2, 0), (0, 3), (3, 1), (2, 3)]         # mid-side nodes 4..9
# 6-node faces and the natural coordinates of their 3 corner + 3 mid-side integration points
_F123, _F023, _F013, _F012 = (1, 2, 3, 5, 8, 9), (0, 2, 3, 6, 7, 9), (0, 1, 3, 4, 7, 8), (0, 1, 2, 4, 5, 6)
_FACE_POINTS = {
    _F123: [[1., 0., 0.], [0., 1., 0.], [0., 0., 0.], [0.5, 0.5, 0.], [0., 0.5, 0.], [0.5, 0., 0.]],
    _F023: [[0., 1., 0.], [0., 0., 1.], [0., 0., 0.], [0., 0.5, 0.], [0., 0.5, 0.5], [0., 0., 0.5]],
    _F013: [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.], [0.5, 0., 0.5], [0., 0.5, 0.5], [0.5, 0.5, 0.]],
    _F012: [[1., 0., 0.], [0., 0., 1.], [0., 0., 0.], [0.5, 0., 0.5], [0.5, 0., 0.], [0., 0., 0.5]],
}
_c, _d, _x = (1. - _A) / (_A - _B), _B / (_A - _B), (0.5 - _B) / (_A - _B)


class Element_quadratic_tetrahedral(ElementBase):
    dm, npe = 3, 10
    _gauss_points = [[_A, _B, _B], [_B, _A, _B], [_B, _B, _A], [_B, _B, _B]]
    _gauss_weights = [1. / 24.] * 4
    facet_natural_coos = _FACE_POINTS
    facet_point_weights = {f: [1. / 12.] * 3 + [1. / 4.] * 3 for f in _FACE_POINTS}
    facet_natural_normals = {f: [TET_FACES[f[:3]]] * 6 for f in _FACE_POINTS}
    inp_surface_num = [(_F012,), (_F013,), (_F123,), (_F023,)]
    _tri_split = [(1, 5, 8), (3, 8, 9), (2, 5, 9), (5, 9, 8), (0, 6, 7), (3, 7, 9), (2, 9, 6), (6, 7, 9),
                  (0, 4, 7), (1, 8, 4), (3, 7, 8), (4, 7, 8), (0, 4, 6), (1, 5, 4), (2, 6, 5), (4, 5, 6)]
    # barycentric coordinates of the 10 nodes w.r.t. the Gauss-point tetrahedron (:321-339)
    _extrap_matrix = np.array([
        [-_d, -_d, 1. + _c, 2. * _d - _c], [1. + _c, -_d, -_d, 2. * _d - _c],
        [-_d, -_d, -_d, 1. + 3. * _d], [-_d, 1. + _c, -_d, 2. * _d - _c],
        [_x, -_d, _x, 1. - 2. * _x + _d], [_x, -_d, -_d, 1. + 2. * _d - _x],
        [-_d, -_d, _x, 1. + 2. * _d - _x], [-_d, _x, _x, 1. - 2. * _x + _d],
        [_x, _x, -_d, 1. - 2. * _x + _d], [-_d, _x, -_d, 1. + 2. * _d - _x]])

    def shapeFunc_pyscope(self, natCoo):
        L = tet_bary(natCoo)
        return np.concatenate([L * (2. * L - 1.), [4. * L[i] * L[j] for i, j in _EDGES]])

    def dshape_dnat_pyscope(self, natCoo):
        L = tet_bary(natCoo)
        corner = (4. * L - 1.)[:, None] * TET_DBARY
        mid = np.array([4. * (L[i] * TET_DBARY[j] + L[j] * TET_DBARY[i]) for i, j in _EDGES])
        return np.concatenate([corner, mid], axis=0)
