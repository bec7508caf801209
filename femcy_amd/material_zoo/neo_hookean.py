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


class NeoHookeanPlaneStrain(MaterBase):
    """the same solid on 2-D elements under plane strain (F33 = 1).  An EXTENSION: the reference has the 3-D class only
    and its reader rejects `*Hyperelastic` on CPS/CPE elements (inp_info.py:296-299), so there is no reference
    result to compare with; BASELINE configs[1] ("CPE8 large-def Neo-Hookean") is what it is for.  Verified against
    the 3-D class on an extruded one-layer mesh with u_z = 0 (tests/test_gpu_tangent.py)."""
    kind = FEMCY_MAT_NEOHOOKE

    def __init__(self, C1: float = 0.4, D1: float = 0.00025):
        self.type, self.dm = "planeStrain", 2
        self.C1, self.D1 = C1, D1
        volume = np.zeros((3, 3))
        volume[:2, :2] = 1.
        self.C = 4. * C1 * np.eye(3) + 2. * D1 * volume        # in-plane rows / columns [xx, yy, xy] of the 3-D C

    @property
    def params(self):
        return np.array([self.C1, self.D1])
