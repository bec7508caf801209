"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/libfemcy_oracle.so (femcy_oracle.c), the
as-written C/OpenMP restatement of the reference's hot kernels.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfemcy_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "femcy_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfemcy_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_cg_solve.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class COracle:
    """reference-layout state for one mesh: ELL sparseIJ / sparseMtrx_rowMajor, ddsdde per Gauss point,
    padded nodeEles -- exactly the fields of System_of_equations.__init__ (stiffnessMtrx.py:33-121)."""

    def __init__(self, nodes, elements, dN, w, C_mat, adj_ptr=None, adj_idx=None):
        self.L = lib()
        self.nodes, self.elements = _f(nodes), _i(elements)
        self.nn, self.dm = self.nodes.shape
        self.ne, self.npe = self.elements.shape
        self.dN, self.w = _f(dN), _f(w)
        self.nGP = self.w.size
        self.s = 3 if self.dm == 2 else 6
        self.n = self.nn * self.dm
        if adj_ptr is None:
            import scipy.sparse as sp
            el = self.elements.astype(np.int64)
            a = np.repeat(el, self.npe, axis=1).ravel()
            b = np.tile(el, (1, self.npe)).ravel()
            adj = sp.coo_matrix((np.ones(a.size, dtype=np.int8), (a, b)), shape=(self.nn, self.nn)).tocsr()
            adj.sum_duplicates()
            adj.sort_indices()
            adj_ptr, adj_idx = adj.indptr, adj.indices
        adj_ptr = np.ascontiguousarray(adj_ptr, dtype=np.int64)
        adj_idx = np.ascontiguousarray(adj_idx, dtype=np.int64)
        self.W = int(np.diff(adj_ptr).max()) * self.dm
        self.ij = np.empty((self.n, self.W + 1), dtype=np.int32)
        self.L.orc_build_sparseIJ(self.nn, self.dm, _p(adj_ptr), _p(adj_idx), self.W, _p(self.ij))
        self.A = np.zeros((self.n, self.W))
        ngp = self.ne * self.nGP
        self.ddsdde = np.empty((ngp, self.s, self.s))
        self.L.orc_ddsdde_init(C.c_int64(ngp), self.s, _p(_f(C_mat)), _p(self.ddsdde))
        self.dsdx = np.zeros((self.ne, self.nGP, self.npe, self.dm))
        self.vol = np.zeros((self.ne, self.nGP))
        self.F = np.zeros((self.ne, self.nGP, self.dm, self.dm))
        self.sigma = np.zeros_like(self.F)
        self._cg_work = None

    def get_dsdx_and_vol(self, dof):
        self.L.orc_get_dsdx_and_vol(self.ne, self.npe, self.dm, self.nGP, _p(self.nodes), _p(_f(dof)), 1,
                                    _p(self.elements), _p(self.dN), _p(self.w), _p(self.dsdx), _p(self.vol))

    def assemble(self):
        self.L.orc_assemble(self.ne, self.npe, self.dm, self.nGP, _p(self.elements), _p(self.dsdx), _p(self.vol),
                            _p(self.ddsdde), _p(self.ij), self.W, C.c_int64(self.n), _p(self.A))

    def internal_force(self, dof, kind, p0, p1):
        dof = _f(dof)
        self.L.orc_deformation_gradient(self.ne, self.npe, self.dm, self.nGP, _p(self.nodes), _p(dof),
                                        _p(self.elements), _p(self.dN), _p(self.F))
        self.L.orc_cauchy_large(C.c_int64(self.ne * self.nGP), self.dm, int(kind), _p(self.ddsdde),
                                C.c_double(p0), C.c_double(p1), _p(self.F), _p(self.sigma))
        self.get_dsdx_and_vol(dof)
        cnt = np.bincount(self.elements.ravel(), minlength=self.nn)
        maxE = int(cnt.max())
        nodeEles = -np.ones((self.nn, maxE), dtype=np.int32)
        order = np.argsort(self.elements.ravel(), kind="stable")
        owners = (order // self.npe).astype(np.int32)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        pos = np.arange(owners.size) - np.repeat(start, cnt)
        nodeEles[np.repeat(np.arange(self.nn), cnt), pos] = owners
        f = np.zeros(self.n)
        self.L.orc_nodal_force(self.nn, self.npe, self.dm, self.nGP, maxE, _p(nodeEles), _p(self.elements),
                               _p(self.dsdx), _p(self.sigma), _p(self.vol), _p(f))
        return f

    def compute_Ad(self, d):
        out = np.empty(self.n)
        self.L.orc_compute_Ad(C.c_int64(self.n), self.W, _p(self.A), _p(self.ij), _p(_f(d)), _p(out))
        return out

    def cg(self, b, eps=1e-3, maxit=0):
        if self._cg_work is None:
            self._cg_work = [np.empty(self.n) for _ in range(5)]
        x, r, d, M, Ad = self._cg_work
        r0, rmax = C.c_double(), C.c_double()
        it = self.L.orc_cg_solve(C.c_int64(self.n), self.W, _p(self.A), _p(self.ij), _p(_f(b)), C.c_double(eps),
                                 int(maxit), _p(x), _p(r), _p(d), _p(M), _p(Ad), C.byref(r0), C.byref(rmax))
        return x.copy(), it, r0.value, rmax.value

    def to_csr(self):
        import scipy.sparse as sp
        cnt = self.ij[:, 0]
        mask = np.arange(self.W)[None, :] < cnt[:, None]
        rows = np.repeat(np.arange(self.n), cnt)
        return sp.coo_matrix((self.A[mask], (rows, self.ij[:, 1:][mask])), shape=(self.n, self.n)).tocsr()

    def zero_rows_cols_unit_diag(self, cons):
        """Dirichlet 0/1 treatment on the ELL arrays (for building a CG test system)."""
        mask = np.zeros(self.n, dtype=bool)
        mask[cons] = True
        valid = np.arange(self.W)[None, :] < self.ij[:, :1]
        cols = np.where(valid, self.ij[:, 1:], 0)
        kill = valid & (mask[:, None] | mask[cols])
        self.A[kill] = 0.0
        diag = valid & (cols == np.arange(self.n)[:, None]) & mask[:, None]
        self.A[diag] = 1.0

    def threads(self):
        return self.L.orc_num_threads()

    def set_threads(self, n: int):
        """OpenMP threads of every later call (process-wide): 1 = serial sums, the reductions exactly as written"""
        self.L.orc_set_num_threads(int(n))
