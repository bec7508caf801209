"""plane-strain linear isotropic material,
cf. /root/reference/material_zoo/linear_isotropic_plane_strain.py:12-42."""
import numpy as np
from .mater_base import MaterBase, FEMCY_MAT_PSTRAIN
from .linear_isotropic import lame_block


class LinearIsotropicPlaneStrain(MaterBase):
    kind = FEMCY_MAT_PSTRAIN

    def __init__(self, modulus: float, poisson_ratio: float):
        self.type, self.dm = "planeStrain", 2
        self.modulus, self.poisson_ratio = modulus, poisson_ratio
        self.G = modulus / 2. / (1. + poisson_ratio)
        t1 = modulus / (1. + poisson_ratio)
        t2 = poisson_ratio / (abs(1. - 2. * poisson_ratio) + 1.e-30)     # guarded as in the reference
        c00, c01 = t1 * (1. + t2), t1 * t2
        self.C = lame_block(c00, c01, self.G, 2, 1)                       # Voigt [xx,yy,xy]
        C6 = lame_block(c00, c01, 0., 3, 3)
        C6[2, 2] = 0.
        C6[3, 3] = self.G
        self.C_6x6 = C6

    @property
    def params(self):
        return np.array([self.modulus, self.poisson_ratio])
