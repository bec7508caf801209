"""element plugins: same class names as the reference's element_zoo/__init__.py:3-8."""
from .element_base import ElementBase, VOIGT_2D, VOIGT_3D
from .element_linear_quadrilateral import Element_linear_quadrilateral
from .element_linear_tetrahedral import Element_linear_tetrahedral
from .element_linear_triangular import Element_linear_triangular
from .element_quadratic_quadrilateral import Element_quadratic_quadrilateral
from .element_quadratic_tetrahedral import Element_quadratic_tetrahedral
from .element_quadratic_triangular import Element_quadratic_triangular

__all__ = ["ElementBase", "VOIGT_2D", "VOIGT_3D", "Element_linear_quadrilateral",
           "Element_linear_tetrahedral", "Element_linear_triangular", "Element_quadratic_quadrilateral",
           "Element_quadratic_tetrahedral", "Element_quadratic_triangular"]
