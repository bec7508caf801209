"""First-principles pins for the four 2-D element families (CPS3/CPE3, CPS4/CPE4, CPS6/CPE6, CPS8/CPE8) and the two
plane materials, derived symbolically (sympy) -- no reference code is executed and none of the hand-written tables of
oracle/elements.py or femcy_amd/element_zoo is an input.  Companion of tests/sympy_pins.py (tetrahedra).

Inputs (SURVEY.md 2b / 9 -- conventions, not code):
  * triangles: natural coordinates (xi, eta); node 0 sits where xi = 1, node 1 where eta = 1, node 2 at the origin
    (`N = [xi, eta, 1 - xi - eta]`, element_linear_triangular.py:62-66); CPS6 mid-side nodes 3, 4, 5 on the edges
    (0,1), (1,2), (2,0);
  * quadrilaterals: natural coordinates in [-1, 1]^2, corners 0..3 counter-clockwise from (-1,-1); CPS8 mid-side
    nodes 4..7 on the edges (0,1), (1,2), (2,3), (3,0) (element_quadratic_quadrilateral.py:7-14);
  * polynomial spaces: P1 / P2 on the triangle, Q1 = span{1, x, y, xy} and the 8-term serendipity space
    span{1, x, y, x^2, xy, y^2, x^2 y, x y^2} on the square;
  * quadrature the reference applies: 1 point (centroid, weight 1/2) for CPS3, the 3 interior points
    (2/3,1/6), (1/6,2/3), (1/6,1/6) with weights 1/6 for CPS6, 2 x 2 Gauss for CPS4 AND for CPS8 (reduced);
  * 2-D Voigt order [xx, yy, xy] with engineering shear; DOF order node * 2 + component; unit thickness;
  * materials: plane strain C(lam, mu); plane stress C = E / (1 - nu^2) [[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]];
    large deformation: plane strain sigma = F S F^T / det F with S = C : E(F) in the plane
    (linear_isotropic_plane_strain.py:76-86); plane stress synthesises F33 = 1 - nu / (1 - nu) (F00 + F11 - 2), forms the
    Green strain of F3 = diag-block(F, F33), applies the PLANE-STRESS C embedded in 6 x 6 (rows and columns of zz, zx, yz
    are zero: linear_isotropic_plane_stress.py:22-31 -- so S33 = 0 and E33 does not enter) and keeps the in-plane part of
    F3 S F3^T / det F3 (:65-96): F33 acts through the determinant only.

Derived here:
  * the shape functions as THE basis of the element's polynomial space that is nodal at its nodes (exact linear solve);
  * K^e = integral of B^T C B over an element with AFFINE geometry (straight-sided triangle, parallelogram), integrated
    exactly in rational arithmetic -- which the reference's rule reproduces for CPS3, CPS6 and CPS4 (asserted by
    the callers through the rule's degree of exactness) -- and, for CPS8, additionally the reference's REDUCED matrix:
    the same exact integrand sampled at the four Gauss points (+-1/sqrt 3 kept symbolic), which differs from the fully
    integrated one;
  * nodal forces, F and Cauchy stress of one element under a homogeneous deformation gradient, in closed form.
"""
from functools import lru_cache

import numpy as np
import sympy as sp

X1, X2 = sp.symbols("xi eta")
NAT = (X1, X2)
R = sp.Rational
KINDS = ("tri3", "tri6", "quad4", "quad8")
ABAQUS = {"CPS3": "tri3", "CPE3": "tri3", "CPS6": "tri6", "CPE6": "tri6", "CPS4": "quad4", "CPE4": "quad4",
          "CPS8": "quad8", "CPE8": "quad8"}


def node_points(kind):
    if kind in ("tri3", "tri6"):
        pts = [(R(1), R(0)), (R(0), R(1)), (R(0), R(0))]
        edges = [(0, 1), (1, 2), (2, 0)]
        quadratic = kind == "tri6"
    else:
        pts = [(R(-1), R(-1)), (R(1), R(-1)), (R(1), R(1)), (R(-1), R(1))]
        edges = [(0, 1), (1, 2), (2, 3), (3, 0)]
        quadratic = kind == "quad8"
    if quadratic:
        pts = pts + [tuple((pts[i][k] + pts[j][k]) / 2 for k in range(2)) for i, j in edges]
    return pts


def monomials(kind):
    return {"tri3": [sp.Integer(1), X1, X2],
            "tri6": [sp.Integer(1), X1, X2, X1 ** 2, X1 * X2, X2 ** 2],
            "quad4": [sp.Integer(1), X1, X2, X1 * X2],
            "quad8": [sp.Integer(1), X1, X2, X1 ** 2, X1 * X2, X2 ** 2, X1 ** 2 * X2, X1 * X2 ** 2]}[kind]


@lru_cache(maxsize=None)
def shape_functions(kind):
    pts, mono = node_points(kind), monomials(kind)
    assert len(pts) == len(mono)
    V = sp.Matrix([[m.subs(dict(zip(NAT, p))) for m in mono] for p in pts])
    coef = V.inv()
    return [sp.expand(sum(coef[k, a] * mono[k] for k in range(len(mono)))) for a in range(len(pts))]


@lru_cache(maxsize=None)
def shape_gradients(kind):
    return [[sp.diff(N, v) for v in NAT] for N in shape_functions(kind)]


def numeric_tables(kind):
    Nf = sp.lambdify(NAT, shape_functions(kind), "numpy")
    dNf = sp.lambdify(NAT, shape_gradients(kind), "numpy")
    return (lambda c: np.array(Nf(*c), dtype=float)), (lambda c: np.array(dNf(*c), dtype=float))


def integrate_ref(kind, expr):
    """exact integral over the reference triangle {xi, eta >= 0, xi + eta <= 1} or the square [-1, 1]^2"""
    expr = sp.expand(expr)
    if kind.startswith("tri"):
        return sp.integrate(sp.expand(sp.integrate(expr, (X2, 0, 1 - X1))), (X1, 0, 1))
    return sp.integrate(sp.integrate(expr, (X2, -1, 1)), (X1, -1, 1))


def gauss_rule(kind):
    """(points, weights) of the rule the reference applies, exact numbers (sqrt(3) symbolic)"""
    if kind == "tri3":
        return [(R(1, 3), R(1, 3))], [R(1, 2)]
    if kind == "tri6":
        return [(R(2, 3), R(1, 6)), (R(1, 6), R(2, 3)), (R(1, 6), R(1, 6))], [R(1, 6)] * 3
    g = 1 / sp.sqrt(3)
    return [(-g, -g), (g, -g), (g, g), (-g, g)], [sp.Integer(1)] * 4


def corner_coordinates(kind):
    """a positively oriented straight-sided triangle / a parallelogram with rational corners (affine geometry map)"""
    if kind.startswith("tri"):
        # node 2 is the origin of the natural frame: x = X2 + xi (X0 - X2) + eta (X1 - X2), det > 0
        return [(R(9, 4), R(1, 3)), (R(2, 3), R(11, 5)), (R(1, 7), R(1, 9))]
    p0, a, b = (R(1, 5), R(1, 3)), (R(7, 3), R(1, 4)), (R(2, 5), R(9, 5))        # p0, p0 + a, p0 + a + b, p0 + b
    return [p0, (p0[0] + a[0], p0[1] + a[1]), (p0[0] + a[0] + b[0], p0[1] + a[1] + b[1]), (p0[0] + b[0], p0[1] + b[1])]


def element_nodes(kind):
    X = [tuple(c) for c in corner_coordinates(kind)]
    if kind == "tri6":
        X += [tuple((X[i][k] + X[j][k]) / 2 for k in range(2)) for i, j in [(0, 1), (1, 2), (2, 0)]]
    if kind == "quad8":
        X += [tuple((X[i][k] + X[j][k]) / 2 for k in range(2)) for i, j in [(0, 1), (1, 2), (2, 3), (3, 0)]]
    return sp.Matrix(X)


def plane_C(material):
    """material = ("pstrain", E, nu) | ("pstress", E, nu) with rational E, nu"""
    kind, E, nu = material
    G = E / (2 * (1 + nu))
    if kind == "pstrain":
        lam = E * nu / ((1 + nu) * (1 - 2 * nu))
        return sp.Matrix([[lam + 2 * G, lam, 0], [lam, lam + 2 * G, 0], [0, 0, G]])
    c = E / (1 - nu ** 2)
    return sp.Matrix([[c, c * nu, 0], [c * nu, c, 0], [0, 0, G]])


def B_matrix(grads):
    npe = len(grads)
    B = sp.zeros(3, 2 * npe)
    for a, (gx, gy) in enumerate(grads):
        B[0, 2 * a] = gx
        B[1, 2 * a + 1] = gy
        B[2, 2 * a], B[2, 2 * a + 1] = gy, gx
    return B


def _spatial_gradients(kind, x):
    """(grads[a] = (dN/dx, dN/dy) as polynomials of the natural coordinates, det J) for node coordinates x (affine)"""
    dN = shape_gradients(kind)
    npe = len(dN)
    J = sp.Matrix(2, 2, lambda i, j: sp.expand(sum(x[a, i] * dN[a][j] for a in range(npe))))
    assert all(e.is_number for e in J), "affine geometry expected (straight-sided triangle / parallelogram)"
    detJ = J.det()
    assert detJ > 0
    Jinv = J.inv()
    grads = [tuple(sp.expand(sum(dN[a][k] * Jinv[k, j] for k in range(2))) for j in range(2)) for a in range(npe)]
    return grads, detJ


@lru_cache(maxsize=None)
def exact_Ke(kind, material=("pstrain", R(7, 2), R(3, 10))):
    """-> (Ke exactly integrated, Ke by the reference's rule on the exact integrand, X, C), float arrays.  For tri3,
    tri6 and quad4 on affine geometry the two coincide (the rule integrates the integrand's degree exactly)."""
    X = element_nodes(kind)
    grads, detJ = _spatial_gradients(kind, X)
    B = B_matrix(grads)
    C = plane_C(material)
    integrand = (B.T * C * B).applyfunc(sp.expand)
    n = B.shape[1]
    Ke = sp.zeros(n, n)
    Kr = sp.zeros(n, n)
    pts, wts = gauss_rule(kind)
    for i in range(n):
        for j in range(i, n):
            e = integrand[i, j]
            Ke[i, j] = Ke[j, i] = integrate_ref(kind, e) * detJ
            Kr[i, j] = Kr[j, i] = sp.nsimplify(sp.expand(sum(w * e.subs({X1: p[0], X2: p[1]}) for p, w in zip(pts, wts)))) * detJ
    f = lambda M: np.array(M.evalf(30).tolist(), dtype=float)
    return f(Ke), f(Kr), f(X), f(C)


def cauchy_large(material, F):
    """closed-form Cauchy stress (2 x 2, exact) of the reference's large-deformation laws for a 2 x 2 rational F"""
    kind, E, nu = material
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    if kind == "pstrain":
        Eg = (F.T * F - sp.eye(2)) / 2
        S = lam * Eg.trace() * sp.eye(2) + 2 * mu * Eg                      # E33 = 0
        return F * S * F.T / F.det()
    F33 = 1 - nu / (1 - nu) * (F[0, 0] + F[1, 1] - 2)
    Eg = (F.T * F - sp.eye(2)) / 2                                          # in-plane block of the Green strain of F3
    C = plane_C(material)
    Sv = C * sp.Matrix([Eg[0, 0], Eg[1, 1], 2 * Eg[0, 1]])                  # S33 = S13 = S23 = 0 by the embedding
    S = sp.Matrix([[Sv[0], Sv[2]], [Sv[2], Sv[1]]])
    return F * S * F.T / (F.det() * F33)


def homogeneous_case(kind, material):
    """one affine element under u = (F - I) X: -> (f [2 npe], u [2 npe], X, F, sigma) as float arrays, the nodal forces
    f_a = sigma . integral over the CURRENT element of grad_x N_a (unit thickness), every integral exact"""
    X = element_nodes(kind)
    npe = X.shape[0]
    F = sp.Matrix([[R(11, 10), R(1, 20)], [R(-1, 25), R(19, 20)]])
    sigma = cauchy_large(material, F)
    x = X * F.T
    grads, detJ = _spatial_gradients(kind, x)
    f = []
    for a in range(npe):
        g = [integrate_ref(kind, grads[a][j]) * detJ for j in range(2)]
        for i in range(2):
            f.append(sum(g[j] * sigma[j, i] for j in range(2)))
    fl = lambda M: np.array(sp.Matrix(M).evalf(30).tolist(), dtype=float)
    return fl(f).ravel(), fl(x - X).ravel(), fl(X), fl(F), fl(sigma)
