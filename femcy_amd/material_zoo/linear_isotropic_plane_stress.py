"""plane-stress linear isotropic material; sigma(F) embeds F in 3-D with a synthesised F33 and
uses C_6x6, cf. /root/reference/material_zoo/linear_isotropic_plane_stress.py:12-34, 65-96."""
import numpy as np
from .mater_base import MaterBase, FEMCY_MAT_PSTRESS
from .linear_isotropic import lame_block


class LinearIsotropicPlaneStress(MaterBase):
    kind = FEMCY_MAT_PSTRESS

    def __init__(self, modulus: float, poisson_ratio: float):
        self.type, self.dm = "planeStress", 2
        self.modulus, self.poisson_ratio = modulus, poisson_ratio
        self.G = modulus / 2. / (1. + poisson_ratio)
        c00 = modulus / (1. - poisson_ratio ** 2)
        c01 = c00 * poisson_ratio
        self.C = lame_block(c00, c01, self.G, 2, 1)
        C6 = np.zeros((6, 6))
        C6[:2, :2] = self.C[:2, :2]
        C6[3, 3] = self.G
        self.C_6x6 = C6

    @property
    def params(self):
        return np.array([self.modulus, self.poisson_ratio])
