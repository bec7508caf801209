"""element plugins.  The package exposes the class names the reference's element_zoo exposes
(element_zoo/__init__.py:3-8) -- `Element_<order>_<shape>`, one module each -- plus the plugin base class."""
from importlib import import_module

from .element_base import ElementBase, VOIGT_2D, VOIGT_3D

__all__ = ["ElementBase", "VOIGT_2D", "VOIGT_3D"]
for _order in ("linear", "quadratic"):
    for _shape in ("triangular", "quadrilateral", "tetrahedral"):
        _name = f"Element_{_order}_{_shape}"
        globals()[_name] = getattr(import_module(f"{__name__}.{_name.lower()}"), _name)
        __all__.append(_name)
del _order, _shape, _name
