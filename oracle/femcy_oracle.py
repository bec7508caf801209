"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the femcy_amd HIP path.

This file is a numpy/scipy *restatement* of the algorithm of mo-hanxuan/FEMcy's solve
path (the reference needs Taichi, which is not installable here, so it can be neither
imported nor run: SURVEY.md 8c).  Nothing in the product package `femcy_amd` may import
it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as
the checker.

PARITY STATUS: "parity unpinned" against a real Taichi run (none can be produced).  The
restatement is pinned instead by the reference's own published known answers -- README.md:66-71,
FEMcy row: sigma_yy at D = 93.56 (CPS3), 93.32 node / 84.40 Gauss point (CPS6), all three
reproduced to the printed digit through the reference's own CG at eps = 1e-3 (round 5:
tests/test_oracle_c.py::test_as_written_cg_reproduces_all_three_published_numbers,
tests/test_oracle_pins.py::test_readme_*); README.md:95 Fig. 2 (d), FEMcy's own large-deformation
load-deflection curve of the cantilever, marker centres measured in the picture (8 points, worst
0.15 of 29.1: test_readme_load_deflection_curve_pins_the_large_deformation_path) and its GIF of the
same beam bending (21 frames, bounding boxes within a pixel: test_reference_gif_of_the_bending_beam_*)
-- and by analytic
properties (tests/test_oracle_*.py).

All `file:line` citations are relative to /root/reference.

Layout conventions (reference): DOF i = node*dm + component (stiffnessMtrx.py:179-180);
3D Voigt order [xx,yy,zz,xy,zx,yz] with engineering shear (linear_isotropic.py:22-31);
2D Voigt [xx,yy,xy]; everything float64 (main.py:11 default_fp=ti.f64).
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sl

try:  # allow "import oracle.femcy_oracle" and "import femcy_oracle"
    from .elements import ElemDef, elem_def
except ImportError:  # pragma: no cover
    from elements import ElemDef, elem_def


# =============================================================================== materials
@dataclass
class Material:
    """kind in {"lin3d","pstrain","pstress","neohooke"}; params: (E, nu) or (C1, D1)."""
    kind: str
    params: Tuple[float, float]

    @property
    def dm(self):
        return 3 if self.kind in ("lin3d", "neohooke") else 2

    @property
    def type(self):  # stiffnessMtrx.py:449-454
        return {"lin3d": "3d", "neohooke": "3d", "pstrain": "planeStrain", "pstress": "planeStress"}[self.kind]

    @property
    def C(self) -> np.ndarray:
        a, b = self.params
        if self.kind == "lin3d":                       # linear_isotropic.py:12-33
            E, nu = a, b
            G = E / 2. / (1. + nu)
            c00 = E * (1. - nu) / (1. + nu) / (1. - 2. * nu)
            c01 = E * nu / (1. + nu) / (1. - 2. * nu)
            C = np.zeros((6, 6))
            C[:3, :3] = c01
            C[0, 0] = C[1, 1] = C[2, 2] = c00
            C[3, 3] = C[4, 4] = C[5, 5] = G
            return C
        if self.kind == "pstrain":                     # linear_isotropic_plane_strain.py:12-29
            E, nu = a, b
            G = E / 2. / (1. + nu)
            term1 = E / (1. + nu)
            term2 = nu / (abs(1. - 2. * nu) + 1.e-30)
            c00 = term1 * (1. + term2)
            c01 = term1 * term2
            return np.array([[c00, c01, 0.], [c01, c00, 0.], [0., 0., G]])
        if self.kind == "pstress":                     # linear_isotropic_plane_stress.py:12-20
            E, nu = a, b
            G = E / 2. / (1. + nu)
            c00 = E / (1. - nu ** 2)
            c01 = c00 * nu
            return np.array([[c00, c01, 0.], [c01, c00, 0.], [0., 0., G]])
        if self.kind == "neohooke":                    # neo_hookean.py:23-42
            C1, D1 = a, b
            vs = np.zeros((6, 6))
            vs[:3, :3] = 1.
            return 4. * C1 * np.eye(6) + 2. * D1 * vs
        raise ValueError(self.kind)

    @property
    def C_6x6(self) -> np.ndarray:
        """3D embedding used by plane stress sigma(F) and the 2D energies."""
        E, nu = self.params
        G = E / 2. / (1. + nu)
        C = np.zeros((6, 6))
        if self.kind == "pstress":                     # linear_isotropic_plane_stress.py:22-31
            c00 = E / (1. - nu ** 2)
            c01 = c00 * nu
            C[0, 0] = C[1, 1] = c00
            C[0, 1] = C[1, 0] = c01
            C[3, 3] = G
            return C
        if self.kind == "pstrain":                     # linear_isotropic_plane_strain.py:31-40
            c = self.C
            c00, c01 = c[0, 0], c[0, 1]
            C[0, 0] = C[1, 1] = c00
            C[0, 1] = C[1, 0] = C[0, 2] = C[2, 0] = C[1, 2] = C[2, 1] = c01
            C[3, 3] = G
            return C
        raise ValueError(self.kind)


def _voigt3(E):
    """[..,3,3] tensor -> [..,6] Voigt with engineering shear, order xx,yy,zz,xy,zx,yz."""
    return np.stack([E[..., 0, 0], E[..., 1, 1], E[..., 2, 2],
                     2. * E[..., 0, 1], 2. * E[..., 2, 0], 2. * E[..., 1, 2]], axis=-1)


def _unvoigt3(v):
    S = np.empty(v.shape[:-1] + (3, 3), dtype=v.dtype)
    S[..., 0, 0], S[..., 1, 1], S[..., 2, 2] = v[..., 0], v[..., 1], v[..., 2]
    S[..., 0, 1] = S[..., 1, 0] = v[..., 3]
    S[..., 0, 2] = S[..., 2, 0] = v[..., 4]
    S[..., 1, 2] = S[..., 2, 1] = v[..., 5]
    return S


def _det(A):
    if A.shape[-1] == 2:
        return A[..., 0, 0] * A[..., 1, 1] - A[..., 0, 1] * A[..., 1, 0]
    return (A[..., 0, 0] * (A[..., 1, 1] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 1])
            - A[..., 0, 1] * (A[..., 1, 0] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 0])
            + A[..., 0, 2] * (A[..., 1, 0] * A[..., 2, 1] - A[..., 1, 1] * A[..., 2, 0]))


def _inv(A):
    """closed-form adjugate/determinant inverse (what ti.Matrix.inverse() does for 2x2/3x3)."""
    d = _det(A)
    out = np.empty_like(A)
    if A.shape[-1] == 2:
        out[..., 0, 0] = A[..., 1, 1]
        out[..., 0, 1] = -A[..., 0, 1]
        out[..., 1, 0] = -A[..., 1, 0]
        out[..., 1, 1] = A[..., 0, 0]
    else:
        for i in range(3):
            for j in range(3):
                i1, i2 = (i + 1) % 3, (i + 2) % 3
                j1, j2 = (j + 1) % 3, (j + 2) % 3
                # cofactor(j,i) -> adjugate(i,j)
                out[..., i, j] = A[..., j1, i1] * A[..., j2, i2] - A[..., j1, i2] * A[..., j2, i1]
    return out / d[..., None, None]


def _F3_plane_stress(F, nu):
    """linear_isotropic_plane_stress.py:72-75: F embedded in 3D with a synthesised F33."""
    F3 = np.zeros(F.shape[:-2] + (3, 3), dtype=F.dtype)
    F3[..., :2, :2] = F
    F3[..., 2, 2] = -nu / (1. - nu) * (F[..., 0, 0] + F[..., 1, 1] - 2.) + 1.
    return F3


def cauchy_large(mat: Material, F: np.ndarray, ddsdde: Optional[np.ndarray] = None) -> np.ndarray:
    """constitutiveOfLargeDeform for F[..., dm, dm] -> cauchy[..., dm, dm].

    lin3d   : linear_isotropic.py:55-76      StVK, sigma = F S F^T / J
    pstrain : linear_isotropic_plane_strain.py:66-86
    pstress : linear_isotropic_plane_stress.py:65-96 (uses C_6x6, not ddsdde)
    neohooke: neo_hookean.py:66-77
    """
    C = mat.C if ddsdde is None else ddsdde
    if mat.kind == "lin3d":
        E = (np.swapaxes(F, -1, -2) @ F - np.eye(3)) / 2.
        pk2 = _unvoigt3(_voigt3(E) @ C.T)
        return F @ pk2 @ np.swapaxes(F, -1, -2) / _det(F)[..., None, None]
    if mat.kind == "pstrain":
        E = (np.swapaxes(F, -1, -2) @ F - np.eye(2)) / 2.
        ev = np.stack([E[..., 0, 0], E[..., 1, 1], E[..., 0, 1] + E[..., 1, 0]], axis=-1)
        v = ev @ C.T
        pk2 = np.empty_like(F)
        pk2[..., 0, 0], pk2[..., 1, 1] = v[..., 0], v[..., 1]
        pk2[..., 0, 1] = pk2[..., 1, 0] = v[..., 2]
        return F @ pk2 @ np.swapaxes(F, -1, -2) / _det(F)[..., None, None]
    if mat.kind == "pstress":
        F3 = _F3_plane_stress(F, mat.params[1])
        E = (np.swapaxes(F3, -1, -2) @ F3 - np.eye(3)) / 2.
        pk2 = _unvoigt3(_voigt3(E) @ mat.C_6x6.T)
        s = F3 @ pk2 @ np.swapaxes(F3, -1, -2) / _det(F3)[..., None, None]
        return s[..., :2, :2]
    if mat.kind == "neohooke":
        C1, D1 = mat.params
        J = _det(F)[..., None, None]
        B = F @ np.swapaxes(F, -1, -2)
        I = np.eye(3)
        return 2. * C1 / J * (B - I) + 2. * D1 * (J - 1.) * I
    raise ValueError(mat.kind)


def cauchy_small(mat: Material, F: np.ndarray) -> np.ndarray:
    """constitutiveOfSmallDeform (post-processing; linear_isotropic*.py, neo_hookean.py:44-60)."""
    if mat.kind == "lin3d":
        E = (F + np.swapaxes(F, -1, -2)) / 2. - np.eye(3)
        return _unvoigt3(_voigt3(E) @ mat.C.T)
    if mat.kind == "pstrain":
        E = (F + np.swapaxes(F, -1, -2)) / 2. - np.eye(2)
        ev = np.stack([E[..., 0, 0], E[..., 1, 1], E[..., 0, 1] + E[..., 1, 0]], axis=-1)
        v = ev @ mat.C.T
        s = np.empty_like(F)
        s[..., 0, 0], s[..., 1, 1] = v[..., 0], v[..., 1]
        s[..., 0, 1] = s[..., 1, 0] = v[..., 2]
        return s
    if mat.kind == "pstress":
        F3 = _F3_plane_stress(F, mat.params[1])
        E = (F3 + np.swapaxes(F3, -1, -2)) / 2. - np.eye(3)
        return _unvoigt3(_voigt3(E) @ mat.C_6x6.T)[..., :2, :2]
    if mat.kind == "neohooke":
        return cauchy_large(mat, F)
    raise ValueError(mat.kind)


def energy_density(mat: Material, F: np.ndarray) -> np.ndarray:
    """elasticEnergyDensity (linear_isotropic.py:78-99 etc., neo_hookean.py:83-89)."""
    if mat.kind == "neohooke":
        C1, D1 = mat.params
        J = _det(F)
        B = F @ np.swapaxes(F, -1, -2)
        return C1 * (np.trace(B, axis1=-2, axis2=-1) - 3. - 2. * np.log(J)) + D1 * (J - 1.) ** 2
    if mat.kind == "lin3d":
        F3, C6 = F, mat.C
    elif mat.kind == "pstress":
        F3, C6 = _F3_plane_stress(F, mat.params[1]), mat.C_6x6
    else:  # pstrain: linear_isotropic_plane_strain.py:88-100
        F3 = np.zeros(F.shape[:-2] + (3, 3), dtype=F.dtype)
        F3[..., :2, :2] = F
        F3[..., 2, 2] = 1.
        C6 = mat.C_6x6
    E = (np.swapaxes(F3, -1, -2) @ F3 - np.eye(3)) / 2.
    ev = _voigt3(E)
    return np.einsum('...i,ij,...j->...', ev, C6, ev) / 2.


# ================================================================== element-level kernels
def strain_mtrx(dsdx: np.ndarray) -> np.ndarray:
    """B(grad N): [..., npe, dm] -> [..., s, npe*dm]  (strainMtrx of every element class,
    e.g. element_linear_tetrahedral.py:138-177, element_linear_triangular.py:124-145)."""
    npe, dm = dsdx.shape[-2:]
    if dm == 2:
        B = np.zeros(dsdx.shape[:-2] + (3, npe * 2))
        B[..., 0, 0::2] = dsdx[..., :, 0]
        B[..., 1, 1::2] = dsdx[..., :, 1]
        B[..., 2, 0::2] = dsdx[..., :, 1]
        B[..., 2, 1::2] = dsdx[..., :, 0]
        return B
    B = np.zeros(dsdx.shape[:-2] + (6, npe * 3))
    B[..., 0, 0::3] = dsdx[..., :, 0]
    B[..., 1, 1::3] = dsdx[..., :, 1]
    B[..., 2, 2::3] = dsdx[..., :, 2]
    B[..., 3, 0::3] = dsdx[..., :, 1]
    B[..., 3, 1::3] = dsdx[..., :, 0]            # gamma_01
    B[..., 4, 0::3] = dsdx[..., :, 2]
    B[..., 4, 2::3] = dsdx[..., :, 0]            # gamma_20
    B[..., 5, 1::3] = dsdx[..., :, 2]
    B[..., 5, 2::3] = dsdx[..., :, 1]            # gamma_12
    return B


def dsdx_and_vol(nodes, elements, dof, ed: ElemDef):
    """get_dsdx_and_vol (stiffnessMtrx.py:132-150): current-configuration grad N and
    det(J)*w per Gauss point.  Returns dsdx[ne,nGP,npe,dm], vol[ne,nGP] (signed)."""
    dm = ed.dm
    x = nodes[elements] + dof.reshape(-1, dm)[elements]          # [ne,npe,dm]
    dNt = ed.dN_table()                                          # [nGP,npe,dm]
    J = np.einsum('eai,gaj->egij', x, dNt)                       # localNodes^T @ dsdn
    dsdx = np.einsum('gak,egkj->egaj', dNt, _inv(J))
    vol = _det(J) * ed.gauss_weights[None, :]
    return dsdx, vol


def deformation_gradient(nodes, elements, dof, ed: ElemDef):
    """get_deformation_gradient (stiffnessMtrx.py:532-556): F = I + u^T (dN (X^T dN)^-1)."""
    dm = ed.dm
    X = nodes[elements]
    U = dof.reshape(-1, dm)[elements]
    dNt = ed.dN_table()
    J0 = np.einsum('eai,gaj->egij', X, dNt)
    dsdX = np.einsum('gak,egkj->egaj', dNt, _inv(J0))
    return np.einsum('eai,egaj->egij', U, dsdX) + np.eye(dm)


def element_stiffness(dsdx, vol, C):
    """per Gauss point B^T C B vol summed over GPs (stiffnessMtrx.py:169-186): [ne,m,m]."""
    B = strain_mtrx(dsdx)                                        # [ne,nGP,s,m]
    CB = np.einsum('pq,egqm->egpm', C, B)
    return np.einsum('egpm,egpn,eg->emn', B, CB, vol)


# ============================================================================ the "Body"
class Topology:
    """body.py:165-234 + stiffnessMtrx.py:70-107 (vectorised; same sets, sorted order)."""

    def __init__(self, nodes: np.ndarray, elements: np.ndarray, ed: ElemDef):
        self.nodes = np.asarray(nodes, dtype=np.float64)
        self.elements = np.asarray(elements, dtype=np.int64)
        self.ed = ed
        self.nn, self.dm = self.nodes.shape
        self.ne, self.npe = self.elements.shape
        self.n = self.nn * self.dm
        el = self.elements
        # node adjacency (coElement_nodes, body.py:182-194), includes the node itself
        a = np.repeat(el, self.npe, axis=1).ravel()
        b = np.tile(el, (1, self.npe)).ravel()
        adj = sp.coo_matrix((np.ones(a.size, dtype=np.int8), (a, b)), shape=(self.nn, self.nn)).tocsr()
        adj.sum_duplicates()
        adj.sort_indices()
        self.adj_ptr, self.adj_idx = adj.indptr.astype(np.int64), adj.indices.astype(np.int64)
        # nodeEles (body.py:165-179), padded with -1 (stiffnessMtrx.py:71-76)
        order = np.argsort(el.ravel(), kind="stable")
        cnt = np.bincount(el.ravel(), minlength=self.nn)
        self.nodeEles_ptr = np.concatenate([[0], np.cumsum(cnt)])
        self.nodeEles_idx = order // self.npe
        self._boundary = None

    # ------------------------------------------------------------------ reference layouts
    def sparseIJ(self) -> np.ndarray:
        """sparseIJ[n, W+1] (stiffnessMtrx.py:78-89): slot 0 = count, then column ids, -1 pad.
        The reference's column order is Python-set iteration order; sorted order is used here
        (order only changes floating-point summation order in SpMV)."""
        dm = self.dm
        maxLen = int(np.diff(self.adj_ptr).max())
        ij = -np.ones((self.n, maxLen * dm + 1), dtype=np.int32)
        for node0 in range(self.nn):
            nb = self.adj_idx[self.adj_ptr[node0]:self.adj_ptr[node0 + 1]]
            js = (nb[:, None] * dm + np.arange(dm)[None, :]).ravel()
            for i in range(dm):
                ij[node0 * dm + i, 1:len(js) + 1] = js
                ij[node0 * dm + i, 0] = len(js)
        return ij

    def scalar_pattern(self):
        """(rows, cols) of every structural non-zero, CSR order."""
        dm = self.dm
        cnt = np.diff(self.adj_ptr)
        brow = np.repeat(np.arange(self.nn), cnt)
        r = (brow[:, None, None] * dm + np.arange(dm)[None, :, None] + np.zeros((1, 1, dm), dtype=np.int64))
        c = (self.adj_idx[:, None, None] * dm + np.zeros((1, dm, 1), dtype=np.int64) + np.arange(dm)[None, None, :])
        return r.ravel(), c.ravel()

    def boundary(self):
        """get_boundary (body.py:197-216): sorted-global-node facet -> owning element."""
        if self._boundary is None:
            facetDic: Dict[Tuple[int, ...], List[int]] = {}
            for facet in self.ed.facet_natural_coos.keys():
                keys = np.sort(self.elements[:, list(facet)], axis=1)
                for iele, k in enumerate(map(tuple, keys.tolist())):
                    facetDic.setdefault(k, []).append(iele)
            self._boundary = {k: v[0] for k, v in facetDic.items() if len(v) == 1}
        return self._boundary


def assemble_K(topo: Topology, dof: np.ndarray, C: np.ndarray) -> sp.csr_matrix:
    """get_dsdx_and_vol + assemble_stiffnessMtrx (stiffnessMtrx.py:132-186) into CSR whose
    structure is the full node-adjacency (x) dm x dm pattern (explicit zeros kept)."""
    ed = topo.ed
    dsdx, vol = dsdx_and_vol(topo.nodes, topo.elements, dof, ed)
    Ke = element_stiffness(dsdx, vol, C)                         # [ne,m,m]
    dm, npe = topo.dm, topo.npe
    gd = (topo.elements[:, :, None] * dm + np.arange(dm)[None, None, :]).reshape(topo.ne, -1)
    I = np.repeat(gd, npe * dm, axis=1).ravel()
    Jc = np.tile(gd, (1, npe * dm)).ravel()
    pr, pc = topo.scalar_pattern()
    K = sp.coo_matrix((np.concatenate([Ke.ravel(), np.zeros(pr.size)]),
                       (np.concatenate([I, pr]), np.concatenate([Jc, pc]))), shape=(topo.n, topo.n)).tocsr()
    K.sort_indices()
    return K


def internal_force(topo: Topology, dof: np.ndarray, mat: Material):
    """assemble_nodal_force_GN (stiffnessMtrx.py:609-644): F at the reference configuration,
    sigma(F), current-configuration dsdx/vol, f[node] = sum_e sum_g dsdx[nid,:] . sigma * vol.
    Returns (f[n], cauchy[ne,nGP,dm,dm], F, dsdx, vol)."""
    ed = topo.ed
    F = deformation_gradient(topo.nodes, topo.elements, dof, ed)
    sig = cauchy_large(mat, F)
    dsdx, vol = dsdx_and_vol(topo.nodes, topo.elements, dof, ed)
    fe = np.einsum('egaj,egji,eg->eai', dsdx, sig, vol)          # dsdx[nid,:] @ sigma -> [i]
    f = np.zeros(topo.n)
    gd = (topo.elements[:, :, None] * topo.dm + np.arange(topo.dm)[None, None, :])
    np.add.at(f, gd.ravel(), fe.ravel())
    return f, sig, F, dsdx, vol


def consistent_tangent(topo: Topology, dof: np.ndarray, mat: Material, h: float = 1.0e-30) -> sp.csr_matrix:
    """d internal_force / d dof, by complex-step differentiation of the restatement above (no reference
    counterpart: the reference's tangent updates are commented out, neo_hookean.py:62-64, 79-81).  The nodal forces
    of an element depend on that element's DOFs only, so one complex perturbation of local DOF k in EVERY element
    at once gives column k of every element tangent: npe*dm evaluations for the whole mesh, exact to rounding (no
    difference quotient, no step-size error).  It shares no formula with the device kernel it is the oracle for
    (kblock_consistent assembles the spatial elasticity tensor + geometric stiffness analytically)."""
    ed, dm = topo.ed, topo.dm
    el = topo.elements
    ne, npe = el.shape
    m = npe * dm
    Xf = topo.nodes[el].reshape(-1, dm)                          # every element owns private copies of its nodes
    elf = np.arange(ne * npe).reshape(ne, npe)
    Ue = np.asarray(dof, dtype=float).reshape(-1, dm)[el].reshape(ne, m)
    Ke = np.empty((ne, m, m))
    for k in range(m):
        Up = Ue.astype(complex)
        Up[:, k] += 1j * h
        w = Up.reshape(-1)
        F = deformation_gradient(Xf, elf, w, ed)
        sig = cauchy_large(mat, F)
        dsdx, vol = dsdx_and_vol(Xf, elf, w, ed)
        fe = np.einsum('egaj,egji,eg->eai', dsdx, sig, vol)
        Ke[:, :, k] = fe.reshape(ne, m).imag / h
    gd = (el[:, :, None] * dm + np.arange(dm)[None, None, :]).reshape(ne, m)
    rows = np.repeat(gd, m, axis=1).ravel()
    cols = np.tile(gd, (1, m)).ravel()
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(topo.n, topo.n)).tocsr()
    K.sum_duplicates()
    return K


# ===================================================================== boundary conditions
def user_dirichletBC(dof, node_set, dm, dm_specified, nodes, time):
    """user_defined/user_api.py:6-30: rotate the node set about z through (40,5,0) by time*pi."""
    pi = 3.141592653589793
    center = np.array([40., 5., 0.])
    angle = time * pi
    rota = np.array([[math.cos(angle), math.sin(angle), 0.],
                     [-math.sin(angle), math.cos(angle), 0.],
                     [0., 0., 1.]])
    X = nodes[node_set]
    new_x = (X - center) @ rota.T + center
    dof[node_set * dm + dm_specified] = (new_x - X)[:, dm_specified]


def dirichlet_dof(dof, bc, dm, nodes, time):
    """dirichletBC_dof (stiffnessMtrx.py:344-366)."""
    ns = np.asarray(bc["node_set"], dtype=np.int64)
    if not bc["user"]:
        dof[ns * dm + bc["dof"]] = bc["val"]
    else:
        user_dirichletBC(dof, ns, dm, bc["dof"], nodes, time)


def _zero_rows_cols_unit_diag(K: sp.csr_matrix, cons: np.ndarray) -> sp.csr_matrix:
    """zero row+col of every constrained DOF and put 1 on the diagonal, keeping structure."""
    K = K.tocsr(copy=True)
    mask = np.zeros(K.shape[0], dtype=bool)
    mask[cons] = True
    rows = np.repeat(np.arange(K.shape[0]), np.diff(K.indptr))
    kill = mask[rows] | mask[K.indices]
    K.data[kill] = 0.
    diag = kill & (rows == K.indices) & mask[rows]
    K.data[diag] = 1.
    return K


def dirichlet_linear(K: sp.csr_matrix, rhs: np.ndarray, bcs: Sequence[dict], dm: int):
    """dirichletBC_linearEquations (stiffnessMtrx.py:279-307), sequential semantics per BC
    block: rhs[j] -= s*K[j][i] (symmetry), rhs[i] = s, zero row i and column i, K[i][i] = 1."""
    rhs = rhs.copy()
    for bc in bcs:
        cons = np.unique(np.asarray(bc["node_set"], dtype=np.int64) * dm + bc["dof"])
        s = np.zeros(K.shape[0])
        s[cons] = bc["val"]
        rhs -= K @ s                     # column i of K times s_i, all i of this block at once
        rhs[cons] = bc["val"]
        K = _zero_rows_cols_unit_diag(K, cons)
    return K, rhs


def dirichlet_newton(K: sp.csr_matrix, residual: np.ndarray, dof: np.ndarray, bcs, dm, nodes, time):
    """dirichletBC_forNewtonMethod (stiffnessMtrx.py:310-341): re-impose dof values, zero the
    residual rows, 0/1 on K."""
    residual = residual.copy()
    allc = []
    for bc in bcs:
        dirichlet_dof(dof, bc, dm, nodes, time)
        cons = np.asarray(bc["node_set"], dtype=np.int64) * dm + bc["dof"]
        residual[cons] = 0.
        allc.append(cons)
    if allc:
        K = _zero_rows_cols_unit_diag(K, np.unique(np.concatenate(allc)))
    return K, residual


def global_normal(ed: ElemDef, nodes_e: np.ndarray, facet: Sequence[int], integPointId: int = 0):
    """globalNormal of every element class (e.g. element_linear_tetrahedral.py:101-134,
    element_linear_triangular.py:88-121)."""
    facet = tuple(sorted(facet))
    natCoo = np.array(ed.facet_natural_coos[facet][integPointId])
    dxdn = nodes_e.T @ ed.dN(natCoo)
    n = np.array(ed.facet_natural_normals[facet][integPointId]) @ np.linalg.inv(dxdn)
    n = n / (np.linalg.norm(n) + 1.e-30)
    if ed.dm == 2:
        area = np.linalg.norm(nodes_e[facet[0]] - nodes_e[facet[1]])
    else:
        area = 0.5 * np.linalg.norm(np.cross(nodes_e[facet[1]] - nodes_e[facet[0]],
                                             nodes_e[facet[2]] - nodes_e[facet[0]]))
    return n, area * ed.facet_point_weights[facet][integPointId]


def neumann_rhs(topo: Topology, load_facets, load_val: float, load_dir=None) -> np.ndarray:
    """neumannBC (stiffnessMtrx.py:369-411).  NB rhs is zero-filled on every call (:384)."""
    ed, dm = topo.ed, topo.dm
    rhs = np.zeros(topo.n)
    boundary = topo.boundary()
    for facet in load_facets:
        ele = boundary[tuple(facet)]
        enodes = topo.elements[ele].tolist()
        localNodes = topo.nodes[topo.elements[ele]]
        localFacet = [enodes.index(i) for i in facet]
        for node0 in facet:
            nid = enodes.index(node0)
            for integId in range(ed.integPointNum_eachFacet):
                normal, axw = global_normal(ed, localNodes, localFacet, integId)
                if load_dir is None or len(load_dir) == 0:
                    flux = load_val * normal * axw
                else:
                    flux = load_val * np.asarray(load_dir) * axw
                natCoo = np.array(ed.facet_natural_coos[tuple(sorted(localFacet))][integId])
                shapeVal = ed.N(natCoo)[nid]
                for i in range(dm):
                    rhs[node0 * dm + i] += flux[i] * shapeVal
    return rhs


# ==================================================================================== PCG
def ell_from_csr(K: sp.csr_matrix, ij: np.ndarray) -> np.ndarray:
    """pack CSR values into the reference's `sparseMtrx_rowMajor` f64[n,W] using sparseIJ."""
    n, W1 = ij.shape
    A = np.zeros((n, W1 - 1))
    Kc = K.tocsr()
    Kc.sort_indices()
    for i in range(n):
        c = ij[i, 0]
        cols = ij[i, 1:c + 1]
        lo, hi = Kc.indptr[i], Kc.indptr[i + 1]
        pos = np.searchsorted(Kc.indices[lo:hi], cols)
        A[i, :c] = Kc.data[lo:hi][pos]
    return A


def pcg_reference(K: sp.csr_matrix, b: np.ndarray, eps: float = 1.0e-3, maxit: Optional[int] = None,
                  history: bool = False):
    """ConjugateGradientSolver_rowMajor.solve (conjugateGradientSolver.py:103-127) with
    M = 1/diag(K) (:48-51), x0 = 0, stop on max|r| < eps * max|r0|.
    Returns (x, iters, r0, rmax[, hist]); iters = number of loop bodies executed."""
    n = b.shape[0]
    M = 1. / K.diagonal()
    x = np.zeros(n)
    r = b.copy()
    d = M * r
    r0 = np.abs(r).max() if n else 0.
    hist = []
    it = 0
    rmax = r0
    for i in range(n if maxit is None else maxit):
        Ad = K @ d
        rMr = np.dot(r * M, r)
        alpha = rMr / np.dot(d, Ad)
        x = x + alpha * d
        r = r - alpha * Ad
        beta = np.dot(r * M, r) / rMr
        d = M * r + beta * d
        rmax = np.abs(r).max()
        it = i + 1
        if history:
            hist.append(rmax)
        if rmax < eps * r0:
            break
    return (x, it, r0, rmax, np.array(hist)) if history else (x, it, r0, rmax)


# ======================================================================== the solver class
def field_norm(f):
    """tiGadgets.py:28-37: sqrt(sum f^2 / N)."""
    return math.sqrt(float(np.dot(f, f)) / f.size)


class OracleSystem:
    """System_of_equations restated (stiffnessMtrx.py:19-822): linear solve and the
    increment / modified-Newton / line-search drivers with the reference's control flow."""

    def __init__(self, nodes, elements, abaqus_type: str, material: Material, geometric_nonlinear: bool,
                 linear_solver: str = "reference", cg_eps: float = 1.0e-3, verbose: bool = False,
                 cg_backend: str = "numpy", cg_threads: Optional[int] = None):
        """cg_backend "c": the CG branch runs in oracle/femcy_oracle.c (`orc_cg_solve`: the reference's ELL arrays,
        thread-per-row product, one pass per kernel) -- same recurrence as `pcg_reference`, fast enough for the
        >= 1e5-DOF systems that take the reference's CG branch (a numpy run of 8e5 iterations would take hours).
        linear_solver "spsolve": every solve is the sparse LU whatever the size (the limit eps -> 0 of the CG branch:
        the yardstick for runs whose eps = 1e-3 iterates differ)."""
        self.ed = elem_def(abaqus_type)
        self.topo = Topology(nodes, elements, self.ed)
        self.dm = self.topo.dm
        self.material = material
        self.C = material.C
        self.geometric_nonlinear = geometric_nonlinear
        n = self.topo.n
        self.rhs = np.zeros(n)
        self.dof = np.zeros(n)
        self.dof_old = np.zeros(n)
        self.du = np.zeros(n)
        self.nodal_force = np.zeros(n)
        self.residual_nodal_force = np.zeros(n)
        self.time0 = self.time1 = 0.
        self.dt = 0.
        self.K = None
        self.linear_solver = linear_solver    # "reference": <1e5 spsolve else CG (stiffnessMtrx.py:272-276)
        self.cg_eps = cg_eps
        self.cg_backend = cg_backend
        self.cg_threads = cg_threads          # C backend: OpenMP threads of the CG (1 = serial sums, the reductions as written)
        self._co = None
        self.verbose = verbose
        self.log: List[dict] = []             # one entry per linear solve / residual evaluation
        self.n_solves = 0
        self.n_assemblies = 0

    # ------------------------------------------------------------------------- pieces
    def assemble_stiffnessMtrx(self):
        self.K = assemble_K(self.topo, self.dof, self.C)
        self.n_assemblies += 1

    def assemble_nodal_force_GN(self):
        self.nodal_force, self.cauchy_stress, self.F, self.dsdx, self.vol = \
            internal_force(self.topo, self.dof, self.material)

    def solve_dof(self):
        """solve_dof/solve_by_scipy/solve_by_CG (stiffnessMtrx.py:219-276)."""
        b = self.rhs if not self.geometric_nonlinear else self.residual_nodal_force
        use_cg = (self.linear_solver == "cg") or (self.linear_solver == "reference" and self.dof.shape[0] >= 1e5)
        assert self.linear_solver in ("cg", "reference", "spsolve")
        if use_cg:
            x, it, r0, rmax = self._cg(b)
            self.log.append({"solve": "cg", "iters": it, "r0": r0, "rmax": rmax, "time1": self.time1})
            if self.verbose:
                print(f"    cg: {it} iterations, r0 = {r0:.6e}, rmax = {rmax:.6e}", flush=True)
        else:
            x = sl.spsolve(self.K.tocsc(), b)
            self.log.append({"solve": "spsolve"})
        self.n_solves += 1
        self.du = np.array(x)
        if not self.geometric_nonlinear:
            self.dof = self.du                      # aliasing as in :246 / :264
        else:
            self.dof = self.dof - self.du
        return self.du

    def _cg(self, b):
        """ConjugateGradientSolver_rowMajor.solve (conjugateGradientSolver.py:103-127), maxit = n."""
        if self.cg_backend != "c":
            return pcg_reference(self.K, b, eps=self.cg_eps)
        try:
            from .c_oracle import COracle
        except ImportError:  # pragma: no cover
            from c_oracle import COracle
        if self._co is None:
            t = self.topo
            self._co = COracle(t.nodes, t.elements, self.ed.dN_table(), self.ed.gauss_weights, self.C,
                               t.adj_ptr, t.adj_idx)
            self._co_mask = np.arange(self._co.W)[None, :] < self._co.ij[:, :1]
        co = self._co
        K = self.K
        # assemble_K keeps the full node-adjacency pattern with sorted columns = the column order of sparseIJ, so the
        # CSR values ARE the ELL rows (`sparseMtrx_rowMajor`, stiffnessMtrx.py:91-94)
        assert K.nnz == int(co.ij[:, 0].sum()), "K lost its structural pattern"
        co.A[self._co_mask] = K.data
        if self.cg_threads:
            keep = co.threads()
            co.set_threads(self.cg_threads)
            try:
                return co.cg(b, eps=self.cg_eps)
            finally:
                co.set_threads(keep)
        return co.cg(b, eps=self.cg_eps)

    def impose_boundary_condition(self, bcs):
        """stiffnessMtrx.py:504-529."""
        for nb in bcs["neumannBCs"]:
            self.rhs = neumann_rhs(self.topo, nb["face_set"], nb["traction"], nb.get("direction"))
        if not self.geometric_nonlinear:
            self.K, self.rhs = dirichlet_linear(self.K, self.rhs, bcs["dirichletBCs"], self.dm)
        else:
            for bc in bcs["dirichletBCs"]:
                dirichlet_dof(self.dof, bc, self.dm, self.topo.nodes, self.time1)

    def _residual_eval(self, bcs):
        self.assemble_nodal_force_GN()
        self.assemble_stiffnessMtrx()
        self.residual_nodal_force = self.nodal_force - self.rhs
        self.K, self.residual_nodal_force = dirichlet_newton(
            self.K, self.residual_nodal_force, self.dof, bcs["dirichletBCs"], self.dm, self.topo.nodes, self.time1)
        return field_norm(self.residual_nodal_force)

    # ------------------------------------------------------------------------ drivers
    def advance_inc(self, bcs):
        """stiffnessMtrx.py:714-822."""
        self.assemble_stiffnessMtrx()                      # :737-738
        self.impose_boundary_condition(bcs)                # :747
        if not self.geometric_nonlinear:
            self.solve_dof()
            return True, 0
        pre_residual = self._residual_eval(bcs)            # :756-759
        if not hasattr(self, "ini_residual"):
            self.ini_residual = pre_residual
        if self.ini_residual < 1.e-9:
            return True, 0   # reference: UnboundLocalError on newton_loop (:822); benign restatement
        newton_loop = -1
        while pre_residual / (self.ini_residual + 1.e-30) >= 0.01:
            newton_loop += 1
            if newton_loop >= 24:
                return False, newton_loop
            du = self.solve_dof()
            residual = self._residual_eval(bcs)
            if np.isnan(residual):
                return False, newton_loop
            if self.verbose:
                print(f"  newton_loop={newton_loop} residual={residual:.6e}")
            # "boost": go further along du while the residual keeps declining (:793-807)
            relax_loop = -1
            relaxation = 1.
            while 0.1 * pre_residual < residual < pre_residual:
                new_residual = residual
                relax_loop += 1
                if relax_loop >= 10:
                    break
                self.dof = self.dof + (-relaxation) * du
                residual = self._residual_eval(bcs)
                if residual > new_residual:
                    self.dof = self.dof + relaxation * du
                    residual = self._residual_eval(bcs)
                    relaxation *= 0.5
            # "damp": residual grew (:810-819)
            relax_loop = -1
            relaxation = 0.5
            while residual > pre_residual:
                relax_loop += 1
                if relax_loop >= 2:
                    break
                self.dof = self.dof + (1. - relaxation) * du
                du *= relaxation                     # tg.field_multiply(du, relaxation), in place
                residual = self._residual_eval(bcs)
            pre_residual = residual
        return True, newton_loop

    def solve(self, time_incs: dict, dirichlet_bc_info: List[dict], neumann_bc_info: List[dict]):
        """stiffnessMtrx.py:647-711."""
        max_inc, min_inc, max_time = time_incs["max_inc"], time_incs["min_inc"], time_incs["max_time"]
        self.dt = time_incs["ini_inc"]
        neumannBCs = copy.deepcopy(neumann_bc_info)
        dirichletBCs = copy.deepcopy(dirichlet_bc_info)
        bcs = {"neumannBCs": neumannBCs, "dirichletBCs": dirichletBCs}
        kinc = -1
        self.increments = []
        while self.time1 < max_time:
            kinc += 1
            self.time1 = min(self.time0 + self.dt, max_time)
            load_ratio = self.time1 / max_time
            for i, nb in enumerate(neumannBCs):
                nb["traction"] = neumann_bc_info[i]["traction"] * load_ratio
            for i, db in enumerate(dirichletBCs):
                db["val"] = dirichlet_bc_info[i]["val"] * load_ratio
            converged, newton_loop = self.advance_inc(bcs)
            self.increments.append({"kinc": kinc, "time1": self.time1, "dt": self.dt,
                                    "converged": converged, "newton_loop": newton_loop})
            if self.verbose:
                print(f"kinc={kinc} time1={self.time1:.5f} dt={self.dt:.5f} conv={converged} nl={newton_loop}")
            if not converged:
                self.time1 = self.time0
                self.dt /= 4.
                self.dof = self.dof_old.copy()
                kinc -= 1
                if self.dt < min_inc:
                    break
                continue
            if newton_loop <= 8:
                self.dt = min(self.dt * 1.5, max_inc)
            self.dof_old = self.dof.copy()
            self.time0 = self.time1
        return self.dof

    # ----------------------------------------------------------------- post-processing
    def compute_strain_stress(self):
        """compute_strain_stress (stiffnessMtrx.py:436-501)."""
        F = deformation_gradient(self.topo.nodes, self.topo.elements, self.dof, self.ed)
        self.F = F
        if not self.geometric_nonlinear:
            self.cauchy_stress = cauchy_small(self.material, F)
        elif not hasattr(self, "cauchy_stress"):
            self.cauchy_stress = cauchy_large(self.material, F)
        s2 = self.cauchy_stress
        s = np.zeros(s2.shape[:-2] + (3, 3))
        s[..., :self.dm, :self.dm] = s2
        if self.material.type == "planeStrain":
            s[..., 2, 2] = self.material.params[1] * (s2[..., 0, 0] + s2[..., 1, 1])
        dev = s - np.eye(3) * (np.trace(s, axis1=-2, axis2=-1) / 3.)[..., None, None]
        self.mises_stress = np.sqrt(1.5 * np.sum(dev * dev, axis=(-2, -1)))
        return self.cauchy_stress

    def extrapolate(self, internal_vals: np.ndarray) -> np.ndarray:
        """ELE.extrapolate: [ne,nGP] -> [ne,npe] patch-wise nodal values."""
        return internal_vals @ self.ed.extrap.T

    def get_elasEng(self):
        """get_elasEng (stiffnessMtrx.py:592-606); vol is whatever get_dsdx_and_vol left."""
        F = deformation_gradient(self.topo.nodes, self.topo.elements, self.dof, self.ed)
        _, vol = dsdx_and_vol(self.topo.nodes, self.topo.elements,
                              self.dof if self.geometric_nonlinear else np.zeros_like(self.dof), self.ed)
        self.elsEng = float(np.sum(energy_density(self.material, F) * vol))
        return self.elsEng
