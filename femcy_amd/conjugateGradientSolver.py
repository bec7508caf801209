"""`ConjugateGradientSolver_rowMajor`, call-compatible with
/root/reference/conjugateGradientSolver.py:8-127 (constructor, re_init(), solve(), .x, .eps).

The reference's `spm`/`sparseIJ` arguments are its ELL value/index fields; here the matrix lives
in the femcy context (blocked SELL-64), so `spm` is the owning
`System_of_equations` (or a `backend.Context`) and `sparseIJ` is ignored.  The recurrence, the
preconditioner M = 1/diag(A), x0 = 0 and the stopping rule max|r| < eps*max|r0| are the
reference's; they run on the device without host round trips -- as ONE persistent launch per solve where the system
fits (`k_pcg_persist`, `k_pcg_small`), as three fused kernels per iteration otherwise.
"""
from . import backend as be


class ConjugateGradientSolver_rowMajor:

    def __init__(self, spm, sparseIJ=None, b=None, eps=1.0e-3):
        self.ctx = spm.ctx if hasattr(spm, "ctx") else spm
        self.b = b
        self.eps = eps
        self.x = self.ctx.vector(be.VEC_X)
        self.maxit = 0               # 0 -> n, as in the reference
        self.iterations = 0
        self.r0 = self.rmax = 0.0

    def re_init(self):
        """x, r, d, M are re-initialised on the device at the start of every solve()."""

    def solve(self):
        self.iterations, self.r0, self.rmax = self.ctx.pcg(self.b.id, self.x.id, eps=self.eps, maxit=self.maxit)
        print("\033[32;1m the initial residual scale is {} \033[0m".format(self.r0))
        print(f"\033[35;1m the {self.iterations - 1}-th loop, norm of residual is {self.rmax} \033[0m")
        return self.iterations
