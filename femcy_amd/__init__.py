"""femcy_amd -- MI355X (gfx950) native solve path behind FEMcy's Python surface.

Layout (only what the hot path needs):
  csrc/                    HIP kernels + the C ABI (include/femcy.h) -> libfemcy_hip.so
  backend.py               ctypes binding; no CPU fallback
  stiffnessMtrx.py         System_of_equations     (host driver, reference control flow)
  conjugateGradientSolver.py  ConjugateGradientSolver_rowMajor
  reader/ element_zoo/ material_zoo/ body.py tiGadgets.py user_defined/ main.py
                           Taichi-free mirrors of the reference's plugin / entry-point surface
  meshgen.py partition.py  synthetic twist-plate meshes and the element partition for N GPUs
"""
__version__ = "0.1.0"
