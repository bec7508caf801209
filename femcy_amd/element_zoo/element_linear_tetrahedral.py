"""C3D4 linear tetrahedron, N = [zeta, xi, 1-xi-eta-zeta, eta], one-point rule (w = 1/6)
(cf. /root/reference/element_zoo/element_linear_tetrahedral.py:27-82)."""
import numpy as np
from .element_base import ElementBase

# corner faces, outward natural normals and the face-centroid natural coordinate
TET_FACES = {(1, 2, 3): [0., 0., -1.], (0, 2, 3): [-1., 0., 0.], (0, 1, 3): [1., 1., 1.], (0, 1, 2): [0., -1., 0.]}
_CENTROID = {(1, 2, 3): [1. / 3., 1. / 3., 0.], (0, 2, 3): [0., 1. / 3., 1. / 3.],
             (0, 1, 3): [1. / 3., 1. / 3., 1. / 3.], (0, 1, 2): [1. / 3., 0., 1. / 3.]}


def tet_bary(c):
    return np.array([c[2], c[0], 1. - c[0] - c[1] - c[2], c[1]])


TET_DBARY = np.array([[0., 0., 1.], [1., 0., 0.], [-1., -1., -1.], [0., 1., 0.]])   # d(bary)/d(xi,eta,zeta)


class Element_linear_tetrahedral(ElementBase):
    dm, npe = 3, 4
    _gauss_points = [[0.25, 0.25, 0.25]]
    _gauss_weights = [1. / 6.]
    facet_natural_coos = {f: [_CENTROID[f]] for f in TET_FACES}
    facet_point_weights = {f: [1.] for f in TET_FACES}
    facet_natural_normals = {f: [n] for f, n in TET_FACES.items()}
    inp_surface_num = [((0, 1, 2),), ((0, 1, 3),), ((1, 2, 3),), ((0, 2, 3),)]
    _tri_split = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]
    _extrap_matrix = np.ones((4, 1))

    def shapeFunc_pyscope(self, natCoo):
        return tet_bary(natCoo)

    def dshape_dnat_pyscope(self, natCoo):
        return TET_DBARY.copy()
