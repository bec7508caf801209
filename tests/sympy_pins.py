"""First-principles pins for the tetrahedral elements, derived symbolically (sympy) -- no reference code is executed
and none of the hand-written tables of oracle/elements.py or femcy_amd/element_zoo is an input.

Inputs (SURVEY.md 2b -- conventions, not code):
  * the node <-> barycentric-coordinate assignment of the reference's tetrahedra: in natural coordinates
    (xi, eta, zeta) node 0 sits where zeta = 1, node 1 where xi = 1, node 2 at the origin, node 3 where eta = 1
    (`N = [zeta, xi, 1 - xi - eta - zeta, eta]`, element_linear_tetrahedral.py:68-71);
  * the Abaqus mid-side ordering of C3D10: node 4..9 on the edges (0,1), (1,2), (2,0), (0,3), (3,1), (2,3);
  * 3-D Voigt order [xx, yy, zz, xy, zx, yz] with engineering shear; DOF order node * 3 + component.

From these the module derives
  * the shape functions as THE polynomial basis of the complete polynomial space (degree 1 / 2) that is nodal at the
    element's nodes (a 4 x 4 / 10 x 10 exact linear solve) -- no closed-form L(2L-1) / 4 L_i L_j is typed in;
  * K^e = integral over the element of B^T C B for an element with affine geometry, integrated EXACTLY (polynomial
    integration over the reference tetrahedron in rational arithmetic);
  * the internal force of the reference's large-deformation formulation (F on the reference configuration,
    St.Venant-Kirchhoff or the reference's neo-Hookean Cauchy stress, current-configuration gradients and volume)
    for a HOMOGENEOUS deformation gradient, where every integral is exact: f_a = sigma . integral grad_x N_a dv.
"""
from functools import lru_cache

import numpy as np
import sympy as sp

XI, ETA, ZETA = sp.symbols("xi eta zeta")
NAT = (XI, ETA, ZETA)
# natural coordinates of the corner nodes, from N = [zeta, xi, 1 - xi - eta - zeta, eta]
CORNERS = [(0, 0, 1), (1, 0, 0), (0, 0, 0), (0, 1, 0)]
C3D10_EDGES = [(0, 1), (1, 2), (2, 0), (0, 3), (3, 1), (2, 3)]


def node_points(etype):
    pts = [tuple(sp.Rational(v) for v in c) for c in CORNERS]
    if etype == "C3D10":
        pts += [tuple((pts[i][k] + pts[j][k]) / 2 for k in range(3)) for i, j in C3D10_EDGES]
    return pts


@lru_cache(maxsize=None)
def shape_functions(etype):
    """nodal basis of P1 (C3D4) / P2 (C3D10) on the reference tetrahedron, solved from N_a(x_b) = delta_ab"""
    pts = node_points(etype)
    mono = [sp.Integer(1), XI, ETA, ZETA]
    if etype == "C3D10":
        mono += [XI ** 2, ETA ** 2, ZETA ** 2, XI * ETA, ETA * ZETA, ZETA * XI]
    assert len(mono) == len(pts)
    V = sp.Matrix([[m.subs(dict(zip(NAT, p))) for m in mono] for p in pts])      # V[b][k] = mono_k(x_b)
    coef = V.inv()                                                             # N_a = sum_k coef[k][a] mono_k
    return [sp.expand(sum(coef[k, a] * mono[k] for k in range(len(mono)))) for a in range(len(pts))]


@lru_cache(maxsize=None)
def shape_gradients(etype):
    return [[sp.diff(N, v) for v in NAT] for N in shape_functions(etype)]


def numeric_tables(etype):
    """(N(nat) -> [npe], dN(nat) -> [npe, 3]) as numpy callables"""
    Nf = sp.lambdify(NAT, shape_functions(etype), "numpy")
    dNf = sp.lambdify(NAT, shape_gradients(etype), "numpy")
    return (lambda c: np.array(Nf(*c), dtype=float)), (lambda c: np.array(dNf(*c), dtype=float))


def integrate_ref_tet(expr):
    """exact integral over the reference tetrahedron {xi, eta, zeta >= 0, xi + eta + zeta <= 1}"""
    expr = sp.expand(expr)
    i1 = sp.integrate(expr, (ZETA, 0, 1 - XI - ETA))
    i2 = sp.integrate(sp.expand(i1), (ETA, 0, 1 - XI))
    return sp.integrate(sp.expand(i2), (XI, 0, 1))


def corner_coordinates():
    """a positively oriented, non-degenerate tetrahedron with rational corners (det dX/dnat > 0 under the map)"""
    R = sp.Rational
    return [(R(1, 2), R(1, 5), R(7, 3)), (R(9, 4), R(1, 3), R(1, 4)), (R(1, 7), R(1, 9), R(1, 6)), (R(2, 3), R(11, 5), R(1, 2))]


def element_nodes(etype):
    X = [tuple(c) for c in corner_coordinates()]
    if etype == "C3D10":
        X += [tuple((X[i][k] + X[j][k]) / 2 for k in range(3)) for i, j in C3D10_EDGES]      # straight edges: affine map
    return sp.Matrix(X)


def isotropic_C(lam, mu):
    C = sp.zeros(6, 6)
    for i in range(3):
        for j in range(3):
            C[i, j] = lam + (2 * mu if i == j else 0)
        C[3 + i, 3 + i] = mu
    return C


def B_matrix(grads):
    """6 x 3 npe strain matrix from spatial gradients grads[a] = (dN/dx, dN/dy, dN/dz); Voigt [xx,yy,zz,xy,zx,yz]"""
    npe = len(grads)
    B = sp.zeros(6, 3 * npe)
    for a, (gx, gy, gz) in enumerate(grads):
        c = 3 * a
        B[0, c + 0] = gx
        B[1, c + 1] = gy
        B[2, c + 2] = gz
        B[3, c + 0], B[3, c + 1] = gy, gx
        B[4, c + 0], B[4, c + 2] = gz, gx
        B[5, c + 1], B[5, c + 2] = gz, gy
    return B


def _lin_coeffs(expr):
    """[c0, c_xi, c_eta, c_zeta] (exact rationals) of a polynomial of degree <= 1 in the natural coordinates"""
    poly = sp.Poly(sp.expand(expr), XI, ETA, ZETA)
    assert poly.total_degree() <= 1
    out = [sp.Integer(0)] * 4
    for (a, b, c), coef in poly.terms():
        out[0 if (a, b, c) == (0, 0, 0) else 1 + (a, b, c).index(1)] = coef
    return out


@lru_cache(maxsize=None)
def exact_Ke(etype, lam=sp.Rational(3, 2), mu=sp.Rational(5, 4)):
    """exact K^e (rational arithmetic throughout) of the affine element `element_nodes(etype)` with isotropic
    C(lam, mu); returns (Ke as float array, X as float array, C as float array).  With affine geometry the spatial
    gradients are polynomials of degree <= 1 in the natural coordinates, so every entry of B^T C B is integrated
    term by term with the exact moments of the reference tetrahedron."""
    X = element_nodes(etype)
    dN = shape_gradients(etype)
    npe = len(dN)
    J = sp.Matrix(3, 3, lambda i, j: sp.expand(sum(X[a, i] * dN[a][j] for a in range(npe))))      # dx_i / dnat_j
    assert all(e.is_number for e in J), "straight-edged element: the geometry map must be affine"
    detJ = J.det()
    assert detJ > 0
    Jinv = J.inv()
    grads = [tuple(sp.expand(sum(dN[a][k] * Jinv[k, j] for k in range(3))) for j in range(3)) for a in range(npe)]
    B = B_matrix(grads)
    C = isotropic_C(lam, mu)
    mono = [sp.Integer(1), XI, ETA, ZETA]
    M = [[integrate_ref_tet(mi * mj) for mj in mono] for mi in mono]                 # exact moments
    n = 3 * npe
    Bc = {(p, i): _lin_coeffs(B[p, i]) for p in range(6) for i in range(n) if B[p, i] != 0}
    Ke = sp.zeros(n, n)
    for i in range(n):
        for j in range(i, n):                                  # upper triangle; symmetric by construction
            tot = sp.Integer(0)
            for p in range(6):
                if (p, i) not in Bc:
                    continue
                for q in range(6):
                    if C[p, q] == 0 or (q, j) not in Bc:
                        continue
                    bi, bj = Bc[(p, i)], Bc[(q, j)]
                    tot += C[p, q] * sum(bi[k] * bj[l] * M[k][l] for k in range(4) for l in range(4)
                                         if bi[k] != 0 and bj[l] != 0)
            Ke[i, j] = Ke[j, i] = tot * detJ
    f = lambda Mx: np.array(Mx.tolist(), dtype=float)
    return f(Ke), f(X), f(C)


def homogeneous_case(etype, material):
    """exact internal force of the affine element under u = (F - I) X for a rational F, in the reference's
    large-deformation formulation.  material: ("stvk", lam, mu) or ("neohooke", C1, D1).  Returns
    (f [3 npe] float, u [3 npe] float, X float, F float, sigma float)."""
    R = sp.Rational
    X = element_nodes(etype)
    npe = X.shape[0]
    F = sp.Matrix([[R(11, 10), R(1, 20), R(-1, 25)], [R(1, 50), R(19, 20), R(3, 40)], [R(-1, 30), R(1, 16), R(21, 20)]])
    Jd = F.det()
    I3 = sp.eye(3)
    if material[0] == "stvk":
        _, lam, mu = material
        E = (F.T * F - I3) / 2
        S = lam * E.trace() * I3 + 2 * mu * E
        sigma = F * S * F.T / Jd
    else:
        _, C1, D1 = material
        sigma = 2 * C1 / Jd * (F * F.T - I3) + 2 * D1 * (Jd - 1) * I3
    dN = shape_gradients(etype)
    x = X * F.T                                                                   # current coordinates
    J = sp.Matrix(3, 3, lambda i, j: sum(x[a, i] * dN[a][j] for a in range(npe))).applyfunc(sp.simplify)
    detJ = J.det()
    Jinv = J.inv()
    f = []
    for a in range(npe):
        g = [integrate_ref_tet(sum(dN[a][k] * Jinv[k, j] for k in range(3)) * detJ) for j in range(3)]   # int grad_x N_a dv
        for i in range(3):
            f.append(sum(g[j] * sigma[j, i] for j in range(3)))
    fl = lambda M: np.array(sp.Matrix(M).tolist(), dtype=float)
    u = fl(x - X).ravel()
    return fl(f).ravel(), u, fl(X), fl(F), fl(sigma)
