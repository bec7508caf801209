"""compressible Neo-Hookean solid, psi = C1 (I1 - 3 - 2 ln J) + D1 (J - 1)^2,
sigma = 2 C1 / J (B - I) + 2 D1 (J - 1) I, constant tangent 4 C1 I6 + 2 D1 (1 x 1),
cf. /root/reference/material_zoo/neo_hookean.py:15-42, 66-77."""
import numpy as np
from .mater_base import MaterBase, FEMCY_MAT_NEOHOOKE


class NeoHookean(MaterBase):
    kind = FEMCY_MAT_NEOHOOKE

    def __init__(self, C1: float = 0.4, D1: float = 0.00025):
        self.type, self.dm = "3d", 3
        self.C1, self.D1 = C1, D1
        self.eye6 = np.eye(6)
        self.volumeStiffness = np.zeros((6, 6))
        self.volumeStiffness[:3, :3] = 1.
        self.C = self.get_C()

    def get_C(self):
        return 4. * self.C1 * self.eye6 + 2. * self.D1 * self.volumeStiffness

    @property
    def params(self):
        return np.array([self.C1, self.D1])
