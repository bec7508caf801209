"""material plugins: same class names as the reference's material_zoo/__init__.py:5-8."""
from .mater_base import (MaterBase, FEMCY_MAT_LIN3D, FEMCY_MAT_PSTRAIN, FEMCY_MAT_PSTRESS, FEMCY_MAT_NEOHOOKE)
from .linear_isotropic import LinearIsotropic
from .linear_isotropic_plane_strain import LinearIsotropicPlaneStrain
from .linear_isotropic_plane_stress import LinearIsotropicPlaneStress
from .neo_hookean import NeoHookean

__all__ = ["MaterBase", "LinearIsotropic", "LinearIsotropicPlaneStrain", "LinearIsotropicPlaneStress", "NeoHookean"]
