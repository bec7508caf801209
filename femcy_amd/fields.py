"""Host-side stand-ins for the Taichi fields the reference's Python surface exposes.

The reference hands `ti.field` objects around (`system.dof.to_numpy()`, `ELE.gaussPoints.shape[0]`,
`field.fill(0.)`, `dof.copy_from(dof_old)`: /root/reference/main.py:31,39,72,
stiffnessMtrx.py:41,384,695).  `HostField` gives a numpy array the same few methods so the
element/material plugin surface and main.py read the same.  Device-resident vectors are
`femcy_amd.backend.DeviceVector`.
"""
import numpy as np


class HostField(np.ndarray):
    """numpy array with the subset of the ti.field API the reference's callers use."""

    def __new__(cls, data, dtype=np.float64):
        return np.array(data, dtype=dtype).view(cls)

    def to_numpy(self):
        return np.array(self)

    def from_numpy(self, arr):
        self[...] = arr

    def copy_from(self, other):
        self[...] = np.asarray(other.to_numpy() if hasattr(other, "to_numpy") else other)
