"""3-D linear isotropic (St.Venant-Kirchhoff under nlgeom) material,
cf. /root/reference/material_zoo/linear_isotropic.py:12-33."""
import numpy as np
from .mater_base import MaterBase, FEMCY_MAT_LIN3D


def lame_block(diag, off, shear, n_normal, n_shear):
    C = np.zeros((n_normal + n_shear,) * 2)
    C[:n_normal, :n_normal] = off
    C[np.arange(n_normal), np.arange(n_normal)] = diag
    i = np.arange(n_normal, n_normal + n_shear)
    C[i, i] = shear
    return C


class LinearIsotropic(MaterBase):
    kind = FEMCY_MAT_LIN3D

    def __init__(self, modulus: float, poisson_ratio: float):
        self.type, self.dm = "3d", 3
        self.modulus, self.poisson_ratio = modulus, poisson_ratio
        nu = poisson_ratio
        self.G = modulus / 2. / (1. + nu)
        c00 = modulus * (1. - nu) / (1. + nu) / (1. - 2. * nu)
        c01 = modulus * nu / (1. + nu) / (1. - 2. * nu)
        self.C = lame_block(c00, c01, self.G, 3, 3)      # Voigt [xx,yy,zz,xy,zx,yz]

    @property
    def params(self):
        return np.array([self.modulus, self.poisson_ratio])
