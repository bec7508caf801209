"""ctypes binding of the C ABI of include/femcy.h.

Two libraries implement it:
  libfemcy_hip.so  the product: hand-written HIP kernels for gfx950 (csrc/).  The default, and the only thing a run on
                   a GPU box ever uses.
  libfemcy_cpu.so  the same ABI on the host (csrc_cpu/, C++ / OpenMP; the arithmetic of an element is the code the HIP
                   kernels run, csrc/element_math.hpp).  Loaded ONLY when FEMCY_BACKEND=cpu is set (or
                   Context(backend="cpu")): there is no silent fallback -- if the HIP library is missing, or no MI355X
                   is visible, construction of a `Context` raises `FemcyError`.  (The CPU restatement of the reference
                   in `oracle/` is something else again: test infrastructure only.)

`import torch` happens before the HIP library is loaded so that libfemcy_hip.so binds to the HIP
runtime (and, for multi-GPU runs, the RCCL) that PyTorch already mapped into the process: the
wheel bundles its own libamdhip64.so/librccl.so with the same SONAMEs as /opt/rocm's, and two
HIP runtimes in one process do not mix.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEMCY_HIP_LIB") or os.path.join(_HERE, "libfemcy_hip.so")   # override: kernel A/B runs
CPU_LIB_PATH = os.environ.get("FEMCY_CPU_LIB") or os.path.join(_HERE, "libfemcy_cpu.so")


def default_backend() -> str:
    """"hip" unless FEMCY_BACKEND=cpu was set explicitly"""
    kind = os.environ.get("FEMCY_BACKEND", "hip").lower()
    if kind not in ("hip", "cpu"):
        raise FemcyError(f"FEMCY_BACKEND={kind!r}: expected 'hip' or 'cpu'")
    return kind

# enum femcy_vec
VEC_DOF, VEC_RHS, VEC_RESIDUAL, VEC_FORCE, VEC_DU, VEC_DOF_OLD, VEC_X, VEC_TMP0, VEC_TMP1 = range(9)
# enum femcy_gpfield
GP_DSDX, GP_VOL, GP_F, GP_SIGMA, GP_STRAIN, GP_MISES, GP_ENERGY = range(7)
# enum femcy_option
OPT_ASSEMBLY, OPT_PCG_POLL, OPT_TIMING, OPT_SPMV_VARIANT, OPT_EW_GRID, OPT_PCG_GRAPH, OPT_SELL_SIGMA = range(7)
OPT_TANGENT = 7          # 0 = the reference's matrix (default), 1 = consistent tangent (extension)
OPT_EXCHANGE = 8         # multi-rank: 0 = packed all-reduce (default), 1 = neighbour send/recv
OPT_PCG_STORAGE_ORDER = 13   # 1 (default) = the single-rank three-launch PCG keeps its vectors in storage order
OPT_PCG_FUSED_UPDATE = 15    # 1 = one vector kernel per iteration in the single-rank three-launch PCG (default 0: measured slower)
OPT_SPMV_FOOTPRINT = 16      # 1 = storage-order product with the wave's x footprint staged in LDS
OPT_DIRECT_MAX_BYTES = 17   # femcy_direct_solve: largest band it may allocate (bytes, default 48 GiB)
OPT_NODE_ORDER = 14      # 0 = caller's numbering, 1 (default) = measured choice among coordinate orders, 2 + k = forced (before build_pattern)
OPT_PCG_PERSIST = 11     # 1 (default) = persistent one-launch PCG (single rank, <= ~7e5 DOF, matrix <= Infinity Cache); 2 = any matrix size
OPT_PCG_PERSIST_MULTI = 12   # 1 (default) = the persistent kernel across ranks once the mailboxes are exchanged and agreed
OPT_PCG_SMALL = 10       # 1 (default) = one persistent launch per solve for systems that fit LDS
TUNE_PERSIST_VARIANT = 109   # persistent PCG variant bits (-1 default; femcy.h)
TUNE_SKIP_OCCUPANCY_CHECK = 111
TUNE_BARRIER_SPIN_LIMIT = 112
TUNE_PERSIST_L2_ROWS = 113
TUNE_ROWS4_TILE = 116       # ROWS4 assembly, round-5 experiment: 1000 GP + LCUT (tile write-out), 0 = off
TUNE_SPMV_WG_PER_XCD = 101  # SpMV workgroups per XCD: 0 auto (512 beyond 512 tasks per XCD, else 256)
TUNE_SPMV_ROT = 119         # SpMV task lists: -1 auto (by the spread of the row lengths), 0 plain, 1..63 rotated rounds, 64 balanced
TUNE_ROWS4_ORDER = 118      # ROWS4 launch order: -1 auto, 0 longest first, 1 Morton / XCD-contiguous
TUNE_PAIRS = 117            # PAIRS assembly knobs (femcy.h)
TUNE_DIRECT_UPDATE = 115    # femcy_direct_solve tile update: -1 auto, 0 VALU, 1 / 2 matrix cores
TUNE_PERSIST_MAX_MB = 114   # persistent PCG: streamed-matrix limit in MiB (0 = none, the default since round 5; 240 = rounds 2-4)
OPT_OVERLAP = 9          # multi-rank, neighbour exchange: 1 (default) = exchange overlapped with the interior product
ASM_GATHER, ASM_ATOMIC, ASM_ROWS, ASM_AUTO, ASM_GATHER_SYM, ASM_GATHER_SYM_ROWSUM, ASM_ROWS2, ASM_ROWS3, ASM_ROWS4, ASM_PAIRS = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9

EXPORTS = [
    "femcy_ctx_create", "femcy_ctx_destroy", "femcy_last_error", "femcy_version", "femcy_set_option", "femcy_sync",
    "femcy_set_mesh", "femcy_set_element", "femcy_set_material", "femcy_build_pattern", "femcy_get_pattern_info",
    "femcy_vec_upload", "femcy_vec_download", "femcy_vec_fill", "femcy_vec_copy", "femcy_vec_scatter",
    "femcy_vec_sub", "femcy_vec_axpy", "femcy_vec_scale", "femcy_vec_norm", "femcy_vec_absmax",
    "femcy_assemble_K", "femcy_internal_force", "femcy_residual_and_K", "femcy_apply_dirichlet_linear", "femcy_apply_dirichlet_newton",
    "femcy_dofset_create", "femcy_dofset_dirichlet_newton", "femcy_dofset_dirichlet_linear", "femcy_dofset_fill",
    "femcy_dofset_scatter", "femcy_loadset_create", "femcy_loadset_neumann", "femcy_spmv", "femcy_pcg", "femcy_compute_strain_stress", "femcy_elastic_energy", "femcy_extrapolate",
    "femcy_get_K_ell", "femcy_get_K_bsr", "femcy_get_gp_field", "femcy_timing",
    "femcy_timing_reset", "femcy_comm_unique_id", "femcy_comm_local_id", "femcy_comm_init", "femcy_comm_info", "femcy_comm_set_neighbours",
    "femcy_comm_tune", "femcy_iface_sum",
    "femcy_probe_stream", "femcy_probe_exchange", "femcy_persist_streamed_bytes",
    "femcy_comm_mailbox_export", "femcy_comm_mailbox_import", "femcy_comm_persist_agree",
    "femcy_comm_shm_id", "femcy_comm_allgather_host", "femcy_get_node_order", "femcy_probe_mailbox", "femcy_probe_spmv",
    "femcy_direct_solve", "femcy_direct_plan",
]


class FemcyError(RuntimeError):
    """a non-zero femcy_status; `.status` carries the code (include/femcy.h)."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


FEMCY_ENUMERIC = -4
FEMCY_ENOMEM = -6


class DirectInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("band_bytes", C.c_int64), ("bandwidth", C.c_int32), ("panels", C.c_int32),
                ("singular_at", C.c_int32), ("negative_pivots", C.c_int32), ("refinements", C.c_int32),
                ("reserved", C.c_int32), ("residual", C.c_double)]


class PatternInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnzb", C.c_int64), ("nnz", C.c_int64), ("max_row_blocks", C.c_int32),
                ("ell_width", C.c_int32), ("stored_blocks", C.c_int64), ("nslices", C.c_int32),
                ("max_node_elems", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("geom_ms", C.c_double), ("geom_launches", C.c_int64),
                ("assemble_ms", C.c_double), ("assemble_launches", C.c_int64),
                ("force_ms", C.c_double), ("force_launches", C.c_int64),
                ("spmv_ms", C.c_double), ("spmv_launches", C.c_int64),
                ("pcg_ms", C.c_double), ("pcg_iters", C.c_int64),
                ("persist_ms", C.c_double), ("persist_launches", C.c_int64), ("persist_iters", C.c_int64),
                ("solves_three", C.c_int64), ("solves_small", C.c_int64), ("solves_persist", C.c_int64),
                ("barrier_timeouts", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_libs = {}


def load_library(require_gpu_runtime: bool = True, kind: Optional[str] = None):
    """dlopen the library of the backend (default: FEMCY_BACKEND, i.e. "hip").  Raises FemcyError when the in-tree
    library has not been built."""
    kind = kind or default_backend()
    if kind in _libs:
        return _libs[kind]
    if kind == "cpu":
        if not os.path.exists(CPU_LIB_PATH):
            raise FemcyError(f"{CPU_LIB_PATH} not found: build it with `bash femcy_amd/csrc_cpu/build.sh`")
        lib = C.CDLL(CPU_LIB_PATH)
        return _bind(lib, kind)
    if not os.path.exists(LIB_PATH):
        raise FemcyError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no silent CPU fallback; FEMCY_BACKEND=cpu selects the host backend explicitly)")
    if require_gpu_runtime:
        import torch  # noqa: F401  (maps PyTorch's HIP runtime first; see module docstring)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    return _bind(lib, kind)


def _bind(lib, kind):
    p, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    cint = C.c_int
    sig = {
        "femcy_ctx_create": [cint, C.POINTER(p)], "femcy_ctx_destroy": [p], "femcy_version": [],
        "femcy_set_option": [p, cint, i64], "femcy_sync": [p],
        "femcy_set_mesh": [p, i32, i32, p, i32, i32, p], "femcy_set_element": [p, i32, p, p, i32],
        "femcy_set_material": [p, i32, p, p, i32], "femcy_build_pattern": [p],
        "femcy_get_pattern_info": [p, C.POINTER(PatternInfo)],
        "femcy_vec_upload": [p, cint, p, i64], "femcy_vec_download": [p, cint, p, i64],
        "femcy_vec_fill": [p, cint, f64], "femcy_vec_copy": [p, cint, cint],
        "femcy_vec_scatter": [p, cint, p, p, i32], "femcy_vec_sub": [p, cint, cint, cint],
        "femcy_vec_axpy": [p, cint, cint, f64, cint], "femcy_vec_scale": [p, cint, f64],
        "femcy_vec_norm": [p, cint, C.POINTER(f64)], "femcy_vec_absmax": [p, cint, C.POINTER(f64)],
        "femcy_assemble_K": [p, cint], "femcy_internal_force": [p, cint, cint], "femcy_residual_and_K": [p, cint, cint],
        "femcy_apply_dirichlet_linear": [p, p, p, i32, cint], "femcy_apply_dirichlet_newton": [p, p, i32, cint],
        "femcy_dofset_create": [p, p, i32, C.POINTER(i32)], "femcy_dofset_dirichlet_newton": [p, i32, cint],
        "femcy_dofset_dirichlet_linear": [p, i32, f64, cint], "femcy_dofset_fill": [p, i32, cint, f64],
        "femcy_dofset_scatter": [p, i32, cint, p],
        "femcy_loadset_create": [p, i32, i32, i32, p, p, p, p, p, i32, p, p, C.POINTER(i32)],
        "femcy_loadset_neumann": [p, i32, f64, p, cint],
        "femcy_spmv": [p, cint, cint],
        "femcy_pcg": [p, cint, cint, f64, i32, C.POINTER(i32), C.POINTER(f64), C.POINTER(f64)],
        "femcy_compute_strain_stress": [p, cint, cint], "femcy_elastic_energy": [p, cint, C.POINTER(f64)],
        "femcy_extrapolate": [p, cint, cint, p, p],
        "femcy_get_K_ell": [p, p, p], "femcy_get_K_bsr": [p, p, p, p], "femcy_get_gp_field": [p, cint, p],
        "femcy_timing": [p, C.POINTER(Timing)], "femcy_timing_reset": [p],
        "femcy_comm_unique_id": [p], "femcy_comm_local_id": [p], "femcy_comm_info": [p, p, p, p],
        "femcy_comm_set_neighbours": [p, i32, p, p, p], "femcy_comm_tune": [p, i32, C.POINTER(i32), p], "femcy_comm_init": [p, i32, i32, p, i32, p, p, i32, p],
        "femcy_iface_sum": [p, cint],
        "femcy_probe_stream": [p, i64, i32, i32, C.POINTER(f64), C.POINTER(i64)],
        "femcy_probe_exchange": [p, i32, i32, C.POINTER(f64)],
        "femcy_persist_streamed_bytes": [p, C.POINTER(i64)],
        "femcy_comm_mailbox_export": [p, p], "femcy_comm_mailbox_import": [p, i32, p],
        "femcy_comm_shm_id": [p, i64], "femcy_comm_allgather_host": [p, p, i32, p],
        "femcy_get_node_order": [p, C.POINTER(i32), p], "femcy_probe_mailbox": [p, i32, C.POINTER(f64)],
        "femcy_probe_spmv": [p, i32, i32, C.POINTER(f64)],
        "femcy_comm_persist_agree": [p, C.POINTER(i32)],
        "femcy_direct_solve": [p, cint, cint, C.POINTER(DirectInfo)],
        "femcy_direct_plan": [p, C.POINTER(DirectInfo)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = cint
    lib.femcy_last_error.argtypes = []
    lib.femcy_last_error.restype = C.c_char_p
    _libs[kind] = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class DeviceVector:
    """handle to one of the context's HBM-resident solver vectors; quacks like the ti.field the
    reference exposes (`to_numpy`, `from_numpy`, `fill`, `copy_from`, `shape`)."""

    def __init__(self, ctx: "Context", vec_id: int):
        self.ctx, self.id = ctx, vec_id

    @property
    def shape(self):
        return (self.ctx.n,)

    def to_numpy(self) -> np.ndarray:
        return self.ctx.download(self.id)

    def from_numpy(self, arr):
        self.ctx.upload(self.id, arr)

    def fill(self, value: float):
        self.ctx._call("femcy_vec_fill", self.id, float(value))

    def copy_from(self, other: "DeviceVector"):
        self.ctx._call("femcy_vec_copy", self.id, other.id)

    def __len__(self):
        return self.ctx.n


class GaussField:
    """handle to a device Gauss-point field (dsdx, vol, F, cauchy stress)."""

    def __init__(self, ctx: "Context", which: int, tail_shape):
        self.ctx, self.which, self._tail = ctx, which, tuple(tail_shape)

    @property
    def shape(self):
        return (self.ctx.ne, self.ctx.nGP)

    def to_numpy(self) -> np.ndarray:
        out = np.empty((self.ctx.ne, self.ctx.nGP) + self._tail, dtype=np.float64)
        self.ctx._call("femcy_get_gp_field", self.which, _ptr(out))
        return out


class Context:
    """one HIP device + stream + all device state of one System_of_equations (backend "cpu": the host)."""

    def __init__(self, device: int = 0, backend: Optional[str] = None):
        self.backend = backend or default_backend()
        self.lib = load_library(kind=self.backend)
        h = C.c_void_p()
        rc = self.lib.femcy_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise FemcyError(f"femcy_ctx_create({device}) -> {rc}: {self.lib.femcy_last_error().decode()}")
        self._h = h
        self.device = device
        self.n = self.nn = self.dm = self.ne = self.npe = self.nGP = 0

    # ------------------------------------------------------------------------------- plumbing
    def _call(self, name, *args):
        rc = getattr(self.lib, name)(self._h, *args)
        if rc != 0:
            raise FemcyError(f"{name} -> {rc}: {self.lib.femcy_last_error().decode()}", status=rc)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.femcy_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt: int, value: int):
        self._call("femcy_set_option", int(opt), int(value))

    def sync(self):
        self._call("femcy_sync")

    # --------------------------------------------------------------------- problem definition
    def set_mesh(self, nodes: np.ndarray, elements: np.ndarray):
        nodes, elements = _f64(nodes), _i32(elements)
        self.nn, self.dm = nodes.shape
        self.ne, self.npe = elements.shape
        self.n = self.nn * self.dm
        self._call("femcy_set_mesh", self.nn, self.dm, _ptr(nodes), self.ne, self.npe, _ptr(elements))

    def set_element(self, ELE):
        """ELE: an element_zoo plugin (anything with .tables())."""
        t = ELE.tables()
        if t["npe"] != self.npe or t["dm"] != self.dm:
            raise FemcyError(f"element plugin (npe={t['npe']}, dm={t['dm']}) does not match mesh "
                             f"(npe={self.npe}, dm={self.dm})")
        self.nGP = int(t["nGP"])
        self._call("femcy_set_element", self.nGP, _ptr(_f64(t["dN"])), _ptr(_f64(t["w"])), int(t["voigt_kind"]))

    def set_material(self, material):
        params = _f64(material.params)
        self._call("femcy_set_material", int(material.kind), _ptr(_f64(material.C)), _ptr(params), int(params.size))

    def build_pattern(self) -> PatternInfo:
        self._call("femcy_build_pattern")
        return self.pattern_info()

    def node_order(self):
        """(order taken by build_pattern: 0 = the caller's numbering, 1 + k = coordinate order k; mean cache lines per
        wavefront gather of every evaluated candidate, the caller's numbering first)"""
        used, lines = C.c_int32(), (C.c_double * 7)()
        self._call("femcy_get_node_order", C.byref(used), C.cast(lines, C.c_void_p))
        return int(used.value), [float(v) for v in lines]

    def pattern_info(self) -> PatternInfo:
        info = PatternInfo()
        self._call("femcy_get_pattern_info", C.byref(info))
        return info

    # --------------------------------------------------------------------------------- vectors
    def vector(self, vec_id: int) -> DeviceVector:
        return DeviceVector(self, vec_id)

    def upload(self, vec_id: int, arr):
        a = _f64(arr).ravel()
        self._call("femcy_vec_upload", int(vec_id), _ptr(a), a.size)

    def download(self, vec_id: int) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        self._call("femcy_vec_download", int(vec_id), _ptr(out), out.size)
        return out

    def scatter(self, vec_id: int, idx, vals):
        idx, vals = _i32(idx).ravel(), _f64(vals).ravel()
        self._call("femcy_vec_scatter", int(vec_id), _ptr(idx), _ptr(vals), idx.size)

    def vec_sub(self, c: int, a: int, b: int):
        self._call("femcy_vec_sub", c, a, b)

    def vec_axpy(self, a: int, b: int, c: float, d: int):
        self._call("femcy_vec_axpy", a, b, float(c), d)

    def vec_scale(self, v: int, s: float):
        self._call("femcy_vec_scale", v, float(s))

    def vec_norm(self, v: int) -> float:
        out = C.c_double()
        self._call("femcy_vec_norm", v, C.byref(out))
        return out.value

    def vec_absmax(self, v: int) -> float:
        out = C.c_double()
        self._call("femcy_vec_absmax", v, C.byref(out))
        return out.value

    # -------------------------------------------------------------------------------- hot path
    def assemble_K(self, u_vec: int = VEC_DOF):
        self._call("femcy_assemble_K", int(u_vec))

    def internal_force(self, u_vec: int = VEC_DOF, f_vec: int = VEC_FORCE):
        self._call("femcy_internal_force", int(u_vec), int(f_vec))

    def residual_and_K(self, u_vec: int = VEC_DOF, f_vec: int = VEC_FORCE):
        """internal force + matrix of one Newton residual evaluation from ONE element pass."""
        self._call("femcy_residual_and_K", int(u_vec), int(f_vec))

    def dirichlet_linear(self, dofs, vals, rhs_vec: int = VEC_RHS):
        dofs, vals = _i32(dofs).ravel(), _f64(vals).ravel()
        self._call("femcy_apply_dirichlet_linear", _ptr(dofs), _ptr(vals), dofs.size, int(rhs_vec))

    def dirichlet_newton(self, dofs, residual_vec: int = VEC_RESIDUAL):
        dofs = _i32(dofs).ravel()
        self._call("femcy_apply_dirichlet_newton", _ptr(dofs), dofs.size, int(residual_vec))

    # device-resident DOF lists of *Boundary blocks (no per-call upload / sync)
    def dofset(self, dofs) -> int:
        dofs = _i32(dofs).ravel()
        out = C.c_int32()
        self._call("femcy_dofset_create", _ptr(dofs), dofs.size, C.byref(out))
        return out.value

    # device-resident *Dsload surfaces
    def loadset(self, ELE, load_elem, load_ft) -> int:
        """load_elem[k]: element owning loaded facet k; load_ft[k]: its facet type (index into ELE.facet_tables())."""
        t = ELE.facet_tables()
        load_elem, load_ft = _i32(load_elem).ravel(), _i32(load_ft).ravel()
        out = C.c_int32()
        self._call("femcy_loadset_create", t["nft"], t["nfn"], t["nip"], _ptr(t["ft_nodes"]), _ptr(t["N"]),
                   _ptr(t["dN"]), _ptr(t["normal"]), _ptr(t["weight"]), load_elem.size, _ptr(load_elem), _ptr(load_ft),
                   C.byref(out))
        return out.value

    def loadset_neumann(self, ls: int, traction: float, direction=None, rhs_vec: int = VEC_RHS):
        d = None if direction is None or len(direction) == 0 else _f64(direction).ravel()
        if d is not None and d.size < self.dm:
            raise FemcyError(f"load direction needs {self.dm} components")
        self._call("femcy_loadset_neumann", int(ls), float(traction), None if d is None else _ptr(d), int(rhs_vec))

    def dofset_dirichlet_newton(self, ds: int, residual_vec: int = VEC_RESIDUAL):
        self._call("femcy_dofset_dirichlet_newton", int(ds), int(residual_vec))

    def dofset_dirichlet_linear(self, ds: int, value: float, rhs_vec: int = VEC_RHS):
        self._call("femcy_dofset_dirichlet_linear", int(ds), float(value), int(rhs_vec))

    def dofset_fill(self, ds: int, vec_id: int, value: float):
        self._call("femcy_dofset_fill", int(ds), int(vec_id), float(value))

    def dofset_scatter(self, ds: int, vec_id: int, vals):
        self._call("femcy_dofset_scatter", int(ds), int(vec_id), _ptr(_f64(vals).ravel()))

    def spmv(self, x_vec: int, y_vec: int):
        self._call("femcy_spmv", int(x_vec), int(y_vec))

    def pcg(self, b_vec: int, x_vec: int = VEC_X, eps: float = 1.0e-3, maxit: int = 0):
        it, r0, rm = C.c_int32(), C.c_double(), C.c_double()
        self._call("femcy_pcg", int(b_vec), int(x_vec), float(eps), int(maxit), C.byref(it), C.byref(r0), C.byref(rm))
        return it.value, r0.value, rm.value

    def direct_solve(self, b_vec: int, x_vec: int = VEC_X) -> dict:
        """vec[x] = K^-1 vec[b] by a band factorisation K = L S L^T (solve_by_scipy, stiffnessMtrx.py:219-251).
        -> what was factored and how good the solution is.  FemcyError with status FEMCY_ENUMERIC when K is singular or
        the residual stays large, FEMCY_ENOMEM when the band is too large."""
        info = DirectInfo()
        self._call("femcy_direct_solve", int(b_vec), int(x_vec), C.byref(info))
        return {k: getattr(info, k) for k, _ in DirectInfo._fields_ if k != "reserved"}

    def direct_plan(self) -> dict:
        """the band `direct_solve` would factor for the current pattern (n, bandwidth, panels, band_bytes), without
        factoring anything"""
        info = DirectInfo()
        self._call("femcy_direct_plan", C.byref(info))
        return {k: getattr(info, k) for k in ("n", "band_bytes", "bandwidth", "panels")}

    # -------------------------------------------------------------------------- post-processing
    def compute_strain_stress(self, u_vec: int = VEC_DOF, large: bool = False):
        self._call("femcy_compute_strain_stress", int(u_vec), int(bool(large)))

    def elastic_energy(self, u_vec: int = VEC_DOF) -> float:
        out = C.c_double()
        self._call("femcy_elastic_energy", int(u_vec), C.byref(out))
        return out.value

    def extrapolate(self, gp_field: int, E: np.ndarray, comp: int = 0) -> np.ndarray:
        """Gauss-point field (component `comp`) -> patch-wise nodal values [ne, npe]; E is npe x nGP."""
        E = _f64(E)
        if E.shape != (self.npe, self.nGP):
            raise FemcyError(f"extrapolation matrix must be {self.npe} x {self.nGP}, got {E.shape}")
        out = np.empty((self.ne, self.npe), dtype=np.float64)
        self._call("femcy_extrapolate", int(gp_field), int(comp), _ptr(E), _ptr(out))
        return out

    # ------------------------------------------------------------------------------ inspection
    def get_K_ell(self):
        info = self.pattern_info()
        W = info.ell_width
        ij = np.empty((self.n, W + 1), dtype=np.int32)
        A = np.empty((self.n, W), dtype=np.float64)
        self._call("femcy_get_K_ell", _ptr(ij), _ptr(A))
        return ij, A

    def get_K_bsr(self):
        """scipy.sparse.bsr_matrix of the device matrix (ascending block columns)."""
        import scipy.sparse as sp
        info = self.pattern_info()
        rowptr = np.empty(self.nn + 1, dtype=np.int32)
        col = np.empty(info.nnzb, dtype=np.int32)
        vals = np.empty((info.nnzb, self.dm, self.dm), dtype=np.float64)
        self._call("femcy_get_K_bsr", _ptr(rowptr), _ptr(col), _ptr(vals))
        return sp.bsr_matrix((vals, col, rowptr), shape=(self.n, self.n))

    def gauss_field(self, which: int) -> GaussField:
        mat = (self.dm, self.dm)
        tail = {GP_DSDX: (self.npe, self.dm), GP_VOL: (), GP_F: mat, GP_SIGMA: mat, GP_STRAIN: mat, GP_MISES: (),
                GP_ENERGY: ()}[which]
        return GaussField(self, which, tail)

    def probe_stream(self, nbytes: int, reps: int = 20, mode: int = 0):
        """(GB/s, bytes per pass) of a read-only sweep in the persistent PCG's launch shape (femcy.h)"""
        us, moved = C.c_double(), C.c_int64()
        self._call("femcy_probe_stream", int(nbytes), int(reps), int(mode), C.byref(us), C.byref(moved))
        return moved.value / (us.value * 1e-6) / 1e9, moved.value

    def probe_exchange(self, rounds: int = 2000, form: int = 0) -> float:
        """microseconds per grid-wide exchange of one value per workgroup (femcy.h)"""
        us = C.c_double()
        self._call("femcy_probe_exchange", int(rounds), int(form), C.byref(us))
        return us.value

    def probe_spmv(self, reps: int = 100, storage_order: bool = True) -> float:
        """microseconds from launch to launch of `reps` back-to-back products (one HIP event pair around the batch)"""
        us = C.c_double()
        self._call("femcy_probe_spmv", int(reps), 1 if storage_order else 0, C.byref(us))
        return float(us.value)

    def probe_mailbox(self, rounds: int = 2000) -> float:
        """collective: microseconds per cross-rank mailbox reduction between the ranks' kernels"""
        us = C.c_double()
        self._call("femcy_probe_mailbox", int(rounds), C.byref(us))
        return float(us.value)

    def persist_streamed_bytes(self) -> int:
        out = C.c_int64()
        self._call("femcy_persist_streamed_bytes", C.byref(out))
        return out.value

    def timing(self) -> dict:
        t = Timing()
        self._call("femcy_timing", C.byref(t))
        return t.as_dict()

    def timing_reset(self):
        self._call("femcy_timing_reset")

    # ------------------------------------------------------------------------------- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.femcy_comm_unique_id(buf)
        if rc != 0:
            raise FemcyError(f"femcy_comm_unique_id -> {rc}: {lib.femcy_last_error().decode()}")
        return buf.raw

    @staticmethod
    def comm_local_id() -> bytes:
        """id of an in-process group: contexts driven by one thread each in this process (single-GPU verification
        of the multi-rank path)."""
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.femcy_comm_local_id(buf)
        if rc != 0:
            raise FemcyError(f"femcy_comm_local_id -> {rc}: {lib.femcy_last_error().decode()}")
        return buf.raw

    @staticmethod
    def comm_shm_id(max_values: int = 1 << 20) -> bytes:
        """id of a shared-memory group: the ranks are processes of one host (any devices, also all on one GPU);
        max_values = the longest vector of one collective, in doubles."""
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.femcy_comm_shm_id(buf, int(max_values))
        if rc != 0:
            raise FemcyError(f"femcy_comm_shm_id -> {rc}: {lib.femcy_last_error().decode()}")
        return buf.raw

    def comm_allgather_host(self, payload: bytes) -> list:
        """collective: every rank's `payload` (same length on all ranks) in rank order, through the context's own
        transport"""
        n = len(payload)
        nranks = C.c_int32()
        self._call("femcy_comm_info", None, C.byref(nranks), None)
        send = C.create_string_buffer(payload, n)
        recv = C.create_string_buffer(n * nranks.value)
        self._call("femcy_comm_allgather_host", send, n, recv)
        return [recv.raw[i * n:(i + 1) * n] for i in range(nranks.value)]

    def comm_mailbox_connect(self) -> bool:
        """export -> all-gather over the context's transport -> import -> agree: True when every rank keeps the
        one-launch PCG"""
        self.comm_mailbox_import(self.comm_allgather_host(self.comm_mailbox_export()))
        return self.comm_persist_agree()

    def comm_init(self, rank: int, nranks: int, uid: bytes, iface_local_dofs, iface_global_slot, niface_global: int,
                  owner):
        d, s = _i32(iface_local_dofs).ravel(), _i32(iface_global_slot).ravel()
        ow = np.ascontiguousarray(owner, dtype=np.uint8).ravel()
        if ow.size != self.n:
            raise FemcyError("owner mask must have one entry per local DOF")
        idbuf = C.create_string_buffer(uid, 128)
        self._call("femcy_comm_init", int(rank), int(nranks), idbuf, d.size, _ptr(d), _ptr(s), int(niface_global),
                   _ptr(ow))
        ng = C.c_int64()
        self._call("femcy_comm_info", None, None, C.byref(ng))
        self.n_global = int(ng.value)

    def comm_info(self):
        """(rank, ranks of the communicator, DOF count of the whole system); (0, 1, n) without a communicator"""
        r, k, ng = C.c_int32(), C.c_int32(), C.c_int64()
        self._call("femcy_comm_info", C.byref(r), C.byref(k), C.byref(ng))
        return int(r.value), int(k.value), int(ng.value)

    def comm_set_neighbours(self, part):
        """part: a `femcy_amd.partition.Part` (nb_ranks / nb_ptr / nb_dofs)."""
        r, q, d = _i32(part.nb_ranks).ravel(), _i32(part.nb_ptr).ravel(), _i32(part.nb_dofs).ravel()
        self._call("femcy_comm_set_neighbours", r.size, _ptr(r), _ptr(q), _ptr(d))

    def comm_tune(self, iters: int = 20) -> dict:
        """collective: measure both interface exchanges, keep the faster one on every rank."""
        chosen, us = C.c_int32(), (C.c_double * 2)()
        self._call("femcy_comm_tune", int(iters), C.byref(chosen), C.cast(us, C.c_void_p))
        return {"exchange": "neighbour" if chosen.value == 1 else "allreduce", "allreduce_us": us[0], "neighbour_us": us[1]}

    # persistent PCG across ranks: the peers' kernels write into each other's mailboxes (femcy.h)
    def comm_mailbox_export(self) -> bytes:
        buf = C.create_string_buffer(256)
        self._call("femcy_comm_mailbox_export", buf)
        return buf.raw

    def comm_mailbox_import(self, blobs) -> None:
        """blobs: the 256-byte exports of ALL ranks, in rank order"""
        joined = b"".join(blobs)
        assert len(joined) == 256 * len(blobs)
        buf = C.create_string_buffer(joined, len(joined))
        self._call("femcy_comm_mailbox_import", len(blobs), buf)

    def comm_persist_agree(self) -> bool:
        """collective: True when every rank can keep the one-launch PCG (femcy_comm_persist_agree)"""
        out = C.c_int32()
        self._call("femcy_comm_persist_agree", C.byref(out))
        return bool(out.value)

    def iface_sum(self, vec_id: int):
        self._call("femcy_iface_sum", int(vec_id))
