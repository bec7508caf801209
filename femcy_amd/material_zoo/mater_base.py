"""Material plugin interface (drop-in for /root/reference/material_zoo/mater_base.py:9-27).

A material owns the constant tangent `C` (the reference copies it to every Gauss point as
`ddsdde` and never updates it, stiffnessMtrx.py:124-129), a `kind` enum + `params` that select
the device constitutive kernel, and the three methods the solver calls.  The constitutive
methods receive device Gauss-point field handles (`femcy_amd.backend.GaussField`) and launch the
HIP kernel of the owning context; there is no host implementation of sigma(F) in the product.
"""
import abc
import numpy as np

# must match include/femcy.h
FEMCY_MAT_LIN3D, FEMCY_MAT_PSTRAIN, FEMCY_MAT_PSTRESS, FEMCY_MAT_NEOHOOKE = 0, 1, 2, 3


class MaterBase(abc.ABC):
    kind: int
    type: str          # "3d" | "planeStrain" | "planeStress"
    dm: int
    C: np.ndarray

    @abc.abstractmethod
    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    @property
    @abc.abstractmethod
    def params(self) -> np.ndarray:
        """numbers the device kernel needs besides C."""

    def _ctx(self, field):
        ctx = getattr(field, "ctx", None)
        if ctx is None:
            raise TypeError("constitutive kernels run on device fields owned by a femcy_amd context; "
                            f"got {type(field).__name__}")
        return ctx

    def constitutiveOfSmallDeform(self, deform_grad, cauchy_stress, ddsdde=None):
        self._ctx(deform_grad).constitutive(large=False)

    def constitutiveOfLargeDeform(self, deform_grad, cauchy_stress, ddsdde=None):
        self._ctx(deform_grad).constitutive(large=True)

    def elasticEnergyDensity(self, deform_grad):
        return self._ctx(deform_grad).energy_density()
