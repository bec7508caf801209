"""material plugins.  The package exposes the class names the reference's material_zoo exposes
(material_zoo/__init__.py:5-8), one module each (module = snake case of the class), plus the plugin base class and
the `kind` codes the C ABI understands."""
import re
from importlib import import_module

from .mater_base import (MaterBase, FEMCY_MAT_LIN3D, FEMCY_MAT_PSTRAIN, FEMCY_MAT_PSTRESS, FEMCY_MAT_NEOHOOKE)

__all__ = ["MaterBase"]
for _name in ("LinearIsotropic", "LinearIsotropicPlaneStrain", "LinearIsotropicPlaneStress", "NeoHookean"):
    _module = "neo_hookean" if _name == "NeoHookean" else re.sub(r"(?<!^)(?=[A-Z])", "_", _name).lower()
    globals()[_name] = getattr(import_module(f"{__name__}.{_module}"), _name)
    __all__.append(_name)
del _name, _module
from .neo_hookean import NeoHookeanPlaneStrain          # extension (no counterpart in the reference)
__all__.append("NeoHookeanPlaneStrain")
