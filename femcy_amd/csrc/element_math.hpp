// The arithmetic of one element / one Gauss point / one facet of the path, defined ONCE for both backends of the C ABI:
// the HIP kernels (kernels_assembly.hip) and the host implementation (csrc_cpu/femcy_cpu.cpp) include this header, so
// the two cannot drift apart.  Everything here is a pure function of its arguments (no memory layout, no threading).
// References (paths relative to the FEMcy checkout) are given at each function.
#pragma once
#include <cmath>
#include <stdint.h>
#include "../../include/femcy.h"

#if defined(__HIPCC__)
#define FEMCY_HD __host__ __device__ __forceinline__
#else
#define FEMCY_HD inline
#endif

namespace femcy {

// ------------------------------------------------------------------------------ small matrices
template <int DM>
FEMCY_HD double det_inv(const double (&J)[DM][DM], double (&inv)[DM][DM]);

template <>
FEMCY_HD double det_inv<2>(const double (&J)[2][2], double (&inv)[2][2]) {
    double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    double id = 1.0 / det;
    inv[0][0] = J[1][1] * id;
    inv[0][1] = -J[0][1] * id;
    inv[1][0] = -J[1][0] * id;
    inv[1][1] = J[0][0] * id;
    return det;
}

template <>
FEMCY_HD double det_inv<3>(const double (&J)[3][3], double (&inv)[3][3]) {
    double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    double id = 1.0 / det;
    inv[0][0] = c00 * id;
    inv[1][0] = c01 * id;
    inv[2][0] = c02 * id;
    inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    return det;
}

FEMCY_HD double det3(const double (&A)[3][3]) {
    return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
           A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}

// sigma = F S F^T / J for a 3x3 F and symmetric S given in Voigt order [xx,yy,zz,xy,zx,yz]
FEMCY_HD void push_forward3(const double (&F)[3][3], const double (&sv)[6], double (&sig)[3][3]) {
    double S[3][3] = {{sv[0], sv[3], sv[4]}, {sv[3], sv[1], sv[5]}, {sv[4], sv[5], sv[2]}};
    double FS[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) FS[i][j] = F[i][0] * S[0][j] + F[i][1] * S[1][j] + F[i][2] * S[2][j];
    double J = det3(F);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sig[i][j] = (FS[i][0] * F[j][0] + FS[i][1] * F[j][1] + FS[i][2] * F[j][2]) / J;
}

// Green strain of a 3x3 F in Voigt order with engineering shear
FEMCY_HD void green_voigt3(const double (&F)[3][3], double (&ev)[6]) {
    double E[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            E[i][j] = (F[0][i] * F[0][j] + F[1][i] * F[1][j] + F[2][i] * F[2][j] - (i == j ? 1.0 : 0.0)) / 2.0;
    ev[0] = E[0][0];
    ev[1] = E[1][1];
    ev[2] = E[2][2];
    ev[3] = 2.0 * E[0][1];
    ev[4] = 2.0 * E[2][0];
    ev[5] = 2.0 * E[1][2];
}

// constitutiveOfLargeDeform: linear_isotropic.py:55-76, linear_isotropic_plane_strain.py:66-86,
// linear_isotropic_plane_stress.py:65-96, neo_hookean.py:66-77.  C is the per-Gauss-point ddsdde
// (a constant copy of material.C in the reference).
template <int DM>
FEMCY_HD void cauchy_large(int kind, const double* __restrict__ C, double p0, double p1,
                                             const double (&F)[DM][DM], double (&sig)[DM][DM]);

template <>
FEMCY_HD void cauchy_large<3>(int kind, const double* __restrict__ C, double p0, double p1,
                                                const double (&F)[3][3], double (&sig)[3][3]) {
    if (kind == FEMCY_MAT_NEOHOOKE) {
        double J = det3(F);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double B = F[i][0] * F[j][0] + F[i][1] * F[j][1] + F[i][2] * F[j][2];
                double eye = (i == j) ? 1.0 : 0.0;
                sig[i][j] = 2.0 * p0 / J * (B - eye) + 2.0 * p1 * (J - 1.0) * eye;
            }
    } else {
        double ev[6], sv[6];
        green_voigt3(F, ev);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) a += C[p * 6 + q] * ev[q];
            sv[p] = a;
        }
        push_forward3(F, sv, sig);
    }
}

template <>
FEMCY_HD void cauchy_large<2>(int kind, const double* __restrict__ C, double p0, double p1,
                                                const double (&F)[2][2], double (&sig)[2][2]) {
    if (kind == FEMCY_MAT_NEOHOOKE) {
        // plane-strain neo-Hookean (extension: the reference has the 3-D form only, neo_hookean.py:66-77, and its
        // reader rejects it on 2-D elements): F33 = 1, so the in-plane part of the 3-D expression with J = det F
        const double J = F[0][0] * F[1][1] - F[0][1] * F[1][0];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double B = F[i][0] * F[j][0] + F[i][1] * F[j][1];
                const double eye = (i == j) ? 1.0 : 0.0;
                sig[i][j] = 2.0 * p0 / J * (B - eye) + 2.0 * p1 * (J - 1.0) * eye;
            }
    } else if (kind == FEMCY_MAT_PSTRESS) {
        // F embedded in 3-D with F33 = 1 - nu/(1-nu) (F00 + F11 - 2); uses C_6x6, not ddsdde
        const double E = p0, nu = p1;
        double F3[3][3] = {{F[0][0], F[0][1], 0.0}, {F[1][0], F[1][1], 0.0}, {0.0, 0.0, 0.0}};
        F3[2][2] = -nu / (1.0 - nu) * (F[0][0] + F[1][1] - 2.0) + 1.0;
        double ev[6], sv[6], s3[3][3];
        green_voigt3(F3, ev);
        const double G = E / 2.0 / (1.0 + nu);
        const double c00 = E / (1.0 - nu * nu), c01 = c00 * nu;
        sv[0] = c00 * ev[0] + c01 * ev[1];
        sv[1] = c01 * ev[0] + c00 * ev[1];
        sv[2] = 0.0;
        sv[3] = G * ev[3];
        sv[4] = 0.0;
        sv[5] = 0.0;
        push_forward3(F3, sv, s3);
        sig[0][0] = s3[0][0];
        sig[0][1] = s3[0][1];
        sig[1][0] = s3[1][0];
        sig[1][1] = s3[1][1];
    } else {
        double E2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) E2[i][j] = (F[0][i] * F[0][j] + F[1][i] * F[1][j] - (i == j ? 1.0 : 0.0)) / 2.0;
        double ev[3] = {E2[0][0], E2[1][1], E2[0][1] + E2[1][0]}, sv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) sv[p] = C[p * 3 + 0] * ev[0] + C[p * 3 + 1] * ev[1] + C[p * 3 + 2] * ev[2];
        double S[2][2] = {{sv[0], sv[2]}, {sv[2], sv[1]}}, FS[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) FS[i][j] = F[i][0] * S[0][j] + F[i][1] * S[1][j];
        double J = F[0][0] * F[1][1] - F[0][1] * F[1][0];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) sig[i][j] = (FS[i][0] * F[j][0] + FS[i][1] * F[j][1]) / J;
    }
}

// ---------------------------------------------------------------------- B_a^T C B_b * vol blocks
template <int DM>
FEMCY_HD void kblock_add(const double* __restrict__ ga, const double* __restrict__ gb,
                                           const double* __restrict__ C, double v, double (&acc)[DM * DM]);

template <>
FEMCY_HD void kblock_add<3>(const double* __restrict__ ga, const double* __restrict__ gb,
                                              const double* __restrict__ C, double v, double (&acc)[9]) {
    const double a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2];
    double CB[6][3];   // C . B_b, non-zeros of B_b only, ascending Voigt index
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        CB[p][0] = C[p * 6 + 0] * b0 + C[p * 6 + 3] * b1 + C[p * 6 + 4] * b2;
        CB[p][1] = C[p * 6 + 1] * b1 + C[p * 6 + 3] * b0 + C[p * 6 + 5] * b2;
        CB[p][2] = C[p * 6 + 2] * b2 + C[p * 6 + 4] * b0 + C[p * 6 + 5] * b1;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        acc[0 * 3 + k] += (a0 * CB[0][k] + a1 * CB[3][k] + a2 * CB[4][k]) * v;
        acc[1 * 3 + k] += (a1 * CB[1][k] + a0 * CB[3][k] + a2 * CB[5][k]) * v;
        acc[2 * 3 + k] += (a2 * CB[2][k] + a0 * CB[4][k] + a1 * CB[5][k]) * v;
    }
}

template <>
FEMCY_HD void kblock_add<2>(const double* __restrict__ ga, const double* __restrict__ gb,
                                              const double* __restrict__ C, double v, double (&acc)[4]) {
    const double a0 = ga[0], a1 = ga[1], b0 = gb[0], b1 = gb[1];
    double CB[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        CB[p][0] = C[p * 3 + 0] * b0 + C[p * 3 + 2] * b1;
        CB[p][1] = C[p * 3 + 1] * b1 + C[p * 3 + 2] * b0;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        acc[0 * 2 + k] += (a0 * CB[0][k] + a1 * CB[2][k]) * v;
        acc[1 * 2 + k] += (a1 * CB[1][k] + a0 * CB[2][k]) * v;
    }
}

// consistent tangent block (FEMCY_OPT_TANGENT = 1), updated-Lagrangian form:
//   K_ab[i][k] = vol * ( gradN_a[j] c_ijkl gradN_b[l]  +  delta_ik gradN_a . sigma . gradN_b )
// with the spatial elasticity tensor c = (1/J) push-forward of dS/dE:
//   StVK (S = C:E, isotropic lambda, mu):  c_ijkl = (lambda b_ij b_kl + mu (b_ik b_jl + b_il b_jk)) / J,  b = F F^T
//   neo-Hookean of neo_hookean.py:66-77 (sigma = 2 C1/J (b - I) + 2 D1 (J-1) I):
//                                          c_ijkl = lambda' d_ij d_kl + mu' (d_ik d_jl + d_il d_jk),
//                                          mu' = 2 C1/J - 2 D1 (J-1),  lambda' = 2 D1 (2J-1)
// checked against central differences of femcy_internal_force (tests/test_gpu_tangent.py).
template <int DM>
FEMCY_HD void kblock_consistent(const double* __restrict__ ga, const double* __restrict__ gb,
                                                  const double* __restrict__ Fp, const double* __restrict__ Sp,
                                                  bool neo, double lam, double mu, double p0, double p1, double v,
                                                  double (&acc)[DM * DM]) {
    double F[DM][DM];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) F[i][j] = Fp[i * DM + j];
    double J;
    if constexpr (DM == 3) J = det3(F);
    else J = F[0][0] * F[1][1] - F[0][1] * F[1][0];
    double geo = 0.0;                       // gradN_a . sigma . gradN_b
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) geo += ga[i] * Sp[i * DM + j] * gb[j];
    if (neo) {
        const double mu_s = 2.0 * p0 / J - 2.0 * p1 * (J - 1.0), lam_s = 2.0 * p1 * (2.0 * J - 1.0);
        double ab = 0.0;
#pragma unroll
        for (int i = 0; i < DM; ++i) ab += ga[i] * gb[i];
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int k = 0; k < DM; ++k)
                acc[i * DM + k] += v * (lam_s * ga[i] * gb[k] + mu_s * gb[i] * ga[k] + (i == k ? mu_s * ab + geo : 0.0));
    } else {
        double b[DM][DM], bga[DM], bgb[DM];
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int j = 0; j < DM; ++j) {
                double t = 0.0;
#pragma unroll
                for (int m = 0; m < DM; ++m) t += F[i][m] * F[j][m];
                b[i][j] = t;
            }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < DM; ++i) {
            double ta = 0.0, tb = 0.0;
#pragma unroll
            for (int j = 0; j < DM; ++j) {
                ta += b[i][j] * ga[j];
                tb += b[i][j] * gb[j];
            }
            bga[i] = ta;
            bgb[i] = tb;
        }
#pragma unroll
        for (int i = 0; i < DM; ++i) s += ga[i] * bgb[i];
        const double vj = v / J;
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int k = 0; k < DM; ++k)
                acc[i * DM + k] += vj * (lam * bga[i] * bgb[k] + mu * (s * b[i][k] + bgb[i] * bga[k])) + (i == k ? v * geo : 0.0);
    }
}

// B_a^T C B_b * v for a C with the cubic sparsity pattern (c11 on the normal diagonal, c12 between normal
// components, c44 on the shear diagonal, zero elsewhere): every material of the reference -- isotropic Hooke
// (c11 = lambda + 2 mu, c12 = lambda, c44 = mu) and the neo-Hookean constant tangent 4 C1 I + 2 D1 1x1 -- has it.
//   K[i][k] = c12 a_i b_k + c44 a_k b_i                     (i != k)
//   K[i][i] = c11 a_i b_i + c44 sum_{j != i} a_j b_j
// 30 multiply-adds instead of the 90 of the dense-pattern evaluation; femcy_set_material detects the pattern with
// exact comparisons and everything else takes kblock_add.
FEMCY_HD void kblock_cubic3(const double* __restrict__ ga, const double* __restrict__ gb, double c11,
                                              double c12, double c44, double v, double (&acc)[9]) {
    const double a0 = ga[0] * v, a1 = ga[1] * v, a2 = ga[2] * v, b0 = gb[0], b1 = gb[1], b2 = gb[2];
    const double p00 = a0 * b0, p11 = a1 * b1, p22 = a2 * b2;
    acc[0] += c11 * p00 + c44 * (p11 + p22);
    acc[4] += c11 * p11 + c44 * (p00 + p22);
    acc[8] += c11 * p22 + c44 * (p00 + p11);
    acc[1] += c12 * (a0 * b1) + c44 * (a1 * b0);
    acc[2] += c12 * (a0 * b2) + c44 * (a2 * b0);
    acc[3] += c12 * (a1 * b0) + c44 * (a0 * b1);
    acc[5] += c12 * (a1 * b2) + c44 * (a2 * b1);
    acc[6] += c12 * (a2 * b0) + c44 * (a0 * b2);
    acc[7] += c12 * (a2 * b1) + c44 * (a1 * b2);
}

// The same block for a material that is the same in every element (c11, c12, c44 are kernel arguments): K_ab is
// linear in the geometric sum S_ab = sum_g |J| w (ga (x) gb), so the row kernels accumulate S over Gauss points AND
// over the incident elements (12 instead of 40 f64 instructions per block and Gauss point) and apply the constants
// once per stored block, when the row is complete: K[i][k] = c12 S[i][k] + c44 S[k][i] (i != k),
// K[i][i] = c11 S[i][i] + c44 (tr S - S[i][i]).
FEMCY_HD void outer3_add(const double* __restrict__ ga, const double* __restrict__ gb, double v, double (&S)[9]) {
    const double a0 = ga[0] * v, a1 = ga[1] * v, a2 = ga[2] * v, b0 = gb[0], b1 = gb[1], b2 = gb[2];
    S[0] += a0 * b0; S[1] += a0 * b1; S[2] += a0 * b2;
    S[3] += a1 * b0; S[4] += a1 * b1; S[5] += a1 * b2;
    S[6] += a2 * b0; S[7] += a2 * b1; S[8] += a2 * b2;
}
FEMCY_HD void cubic_from_outer3(const double (&S)[9], double c11, double c12, double c44, double (&K)[9]) {
    K[0] = c11 * S[0] + c44 * (S[4] + S[8]);
    K[4] = c11 * S[4] + c44 * (S[0] + S[8]);
    K[8] = c11 * S[8] + c44 * (S[0] + S[4]);
    K[1] = c12 * S[1] + c44 * S[3];
    K[3] = c12 * S[3] + c44 * S[1];
    K[2] = c12 * S[2] + c44 * S[6];
    K[6] = c12 * S[6] + c44 * S[2];
    K[5] = c12 * S[5] + c44 * S[7];
    K[7] = c12 * S[7] + c44 * S[5];
}

// ------------------------------------------------------------------------------ post-processing
// compute_strain_stress (stiffnessMtrx.py:436-501), constitutiveOfSmallDeform x4, the three Mises kernels,
// elasticEnergyDensity x4 (material_zoo/*.py), get_elasEng_kernel (:597-606).  One thread per Gauss point.
template <int DM>
FEMCY_HD void cauchy_small(int kind, const double* __restrict__ C, double p0, double p1,
                                             const double (&F)[DM][DM], double (&sig)[DM][DM]);

template <>
FEMCY_HD void cauchy_small<3>(int kind, const double* __restrict__ C, double p0, double p1,
                                                const double (&F)[3][3], double (&sig)[3][3]) {
    if (kind == FEMCY_MAT_NEOHOOKE) {
        cauchy_large<3>(kind, C, p0, p1, F, sig);   // neo_hookean.py:44-60: same expression
        return;
    }
    double E[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[i][j] = (F[i][j] + F[j][i]) / 2.0 - (i == j ? 1.0 : 0.0);
    const double ev[6] = {E[0][0], E[1][1], E[2][2], 2.0 * E[0][1], 2.0 * E[2][0], 2.0 * E[1][2]};
    double sv[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) a += C[p * 6 + q] * ev[q];
        sv[p] = a;
    }
    sig[0][0] = sv[0]; sig[1][1] = sv[1]; sig[2][2] = sv[2];
    sig[0][1] = sig[1][0] = sv[3];
    sig[0][2] = sig[2][0] = sv[4];
    sig[1][2] = sig[2][1] = sv[5];
}

template <>
FEMCY_HD void cauchy_small<2>(int kind, const double* __restrict__ C, double p0, double p1,
                                                const double (&F)[2][2], double (&sig)[2][2]) {
    if (kind == FEMCY_MAT_NEOHOOKE) {
        cauchy_large<2>(kind, C, p0, p1, F, sig);   // as in 3-D: the same expression for small and large deformation
    } else if (kind == FEMCY_MAT_PSTRESS) {
        const double E_ = p0, nu = p1;
        const double F33 = -nu / (1.0 - nu) * (F[0][0] + F[1][1] - 2.0) + 1.0;   // E33 = F33 - 1 multiplies zeros of C_6x6
        (void)F33;
        const double e00 = F[0][0] - 1.0, e11 = F[1][1] - 1.0, g01 = 2.0 * ((F[0][1] + F[1][0]) / 2.0);
        const double G = E_ / 2.0 / (1.0 + nu), c00 = E_ / (1.0 - nu * nu), c01 = c00 * nu;
        sig[0][0] = c00 * e00 + c01 * e11;
        sig[1][1] = c01 * e00 + c00 * e11;
        sig[0][1] = sig[1][0] = G * g01;
    } else {
        const double e00 = F[0][0] - 1.0, e11 = F[1][1] - 1.0;
        const double e01 = (F[0][1] + F[1][0]) / 2.0;
        const double ev[3] = {e00, e11, e01 + e01};
        double v[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) v[p] = C[p * 3 + 0] * ev[0] + C[p * 3 + 1] * ev[1] + C[p * 3 + 2] * ev[2];
        sig[0][0] = v[0];
        sig[1][1] = v[1];
        sig[0][1] = sig[1][0] = v[2];
    }
}

FEMCY_HD double energy_voigt3(const double (&F3)[3][3], const double (&C6)[6][6]) {
    double ev[6];
    green_voigt3(F3, ev);
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) a += C6[p][q] * ev[q];
        acc += ev[p] * a;
    }
    return acc / 2.0;
}

template <int DM>
FEMCY_HD double energy_density(int kind, const double* __restrict__ C, double p0, double p1,
                                                 const double (&F)[DM][DM]) {
    double F3[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, C6[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) C6[i][j] = 0.0;
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) F3[i][j] = F[i][j];
    if (kind == FEMCY_MAT_NEOHOOKE) {
        if (DM == 2) F3[2][2] = 1.0;                 // plane strain
        const double J = det3(F3);
        double trB = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) trB += F3[i][j] * F3[i][j];
        return p0 * (trB - 3.0 - 2.0 * log(J)) + p1 * (J - 1.0) * (J - 1.0);
    }
    if (kind == FEMCY_MAT_LIN3D) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) C6[i][j] = C[i * 6 + j];
    } else if (kind == FEMCY_MAT_PSTRESS) {       // linear_isotropic_plane_stress.py:22-31, 98-114
        const double E_ = p0, nu = p1, c00 = E_ / (1.0 - nu * nu);
        F3[2][2] = -nu / (1.0 - nu) * (F[0][0] + F[1][1] - 2.0) + 1.0;
        C6[0][0] = C6[1][1] = c00;
        C6[0][1] = C6[1][0] = c00 * nu;
        C6[3][3] = E_ / 2.0 / (1.0 + nu);
    } else {                                      // plane strain: linear_isotropic_plane_strain.py:31-40, 88-100
        F3[2][2] = 1.0;
        const double c00 = C[0], c01 = C[1];
        C6[0][0] = C6[1][1] = c00;
        C6[0][1] = C6[1][0] = C6[0][2] = C6[2][0] = C6[1][2] = C6[2][1] = c01;
        C6[3][3] = C[8];
    }
    return energy_voigt3(F3, C6);
}

// one Gauss point of compute_strain_stress (stiffnessMtrx.py:436-501): strain (infinitesimal, or Green when `large`),
// Cauchy stress by constitutiveOfSmallDeform when !large (kept as given otherwise: "stress has been computed for
// geometric nonlinear case", :446-447), von Mises stress by material type (:449-501)
template <int DM>
FEMCY_HD void post_point(int large, int kind, const double* __restrict__ C, double p0, double p1,
                         const double* __restrict__ Fin, double* __restrict__ sigma, double* __restrict__ strain,
                         double* __restrict__ mises) {
    double F[DM][DM], sig[DM][DM];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) F[i][j] = Fin[i * DM + j];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) {
            double e;
            if (large) {
                e = 0.0;
#pragma unroll
                for (int k = 0; k < DM; ++k) e += F[k][i] * F[k][j];
                e = (e - (i == j ? 1.0 : 0.0)) / 2.0;       // Green strain (:575-589)
            } else {
                e = (F[i][j] + F[j][i]) / 2.0 - (i == j ? 1.0 : 0.0);   // infinitesimal strain (:559-572)
            }
            strain[i * DM + j] = e;
        }
    if (!large) {
        cauchy_small<DM>(kind, C, p0, p1, F, sig);
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int j = 0; j < DM; ++j) sigma[i * DM + j] = sig[i][j];
    } else {
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int j = 0; j < DM; ++j) sig[i][j] = sigma[i * DM + j];
    }
    double s3[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) s3[i][j] = sig[i][j];
    if (kind == FEMCY_MAT_PSTRAIN) s3[2][2] = p1 * (sig[0][0] + sig[1][1]);   // nu (s_xx + s_yy) (:475-489)
    if (DM == 2 && kind == FEMCY_MAT_NEOHOOKE)      // plane-strain neo-Hookean: b33 = 1 leaves the volumetric part
        s3[2][2] = 2.0 * p1 * (F[0][0] * F[1][1] - F[0][1] * F[1][0] - 1.0);
    const double tr = (s3[0][0] + s3[1][1] + s3[2][2]) / 3.0;
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double dv = s3[i][j] - (i == j ? tr : 0.0);
            ss += dv * dv;
        }
    *mises = sqrt(1.5 * ss);
}

// one loaded facet of neumannBC (stiffnessMtrx.py:369-411, ELE.globalNormal): the loads of the facet's nodes on the
// UNDEFORMED element, contrib[fn][d]; `en` = the owning element's nodes, `t` = facet type (index into the element
// plugin's facet tables), dir_or_null = traction direction (TRVEC) or NULL for the outward normal (pressure)
template <int DM>
FEMCY_HD void neumann_facet(int32_t npe, int32_t nfn, int32_t nip, const double* __restrict__ nodes,
                            const int32_t* __restrict__ en, int32_t t, const int32_t* __restrict__ ft_nodes,
                            const double* __restrict__ ft_N, const double* __restrict__ ft_dN,
                            const double* __restrict__ ft_normal, const double* __restrict__ ft_weight, double traction,
                            const double* __restrict__ dir_or_null, double* __restrict__ contrib) {
    const int32_t* key = ft_nodes + (int64_t)t * nfn;
    // facet size from the first sorted local nodes (ELE.globalNormal): edge length, or corner-triangle area
    double size;
    {
        const double* p0 = nodes + (int64_t)en[key[0]] * DM;
        const double* p1 = nodes + (int64_t)en[key[1]] * DM;
        if (DM == 2) {
            const double dx = p0[0] - p1[0], dy = p0[1] - p1[1];
            size = sqrt(dx * dx + dy * dy);
        } else {
            const double* p2 = nodes + (int64_t)en[key[2]] * DM;
            const double a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
            const double b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
            const double c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
            size = 0.5 * sqrt(c0 * c0 + c1 * c1 + c2 * c2);
        }
    }
    for (int32_t fn = 0; fn < nfn; ++fn)
        for (int d = 0; d < DM; ++d) contrib[(int64_t)fn * DM + d] = 0.0;
    for (int32_t ip = 0; ip < nip; ++ip) {
        const int64_t tip = (int64_t)t * nip + ip;
        double J[DM][DM], inv[DM][DM];
        for (int i = 0; i < DM; ++i)
            for (int j = 0; j < DM; ++j) J[i][j] = 0.0;
        for (int32_t a = 0; a < npe; ++a) {
            const double* x = nodes + (int64_t)en[a] * DM;
            const double* dn = ft_dN + (tip * npe + a) * DM;
            for (int i = 0; i < DM; ++i)
                for (int j = 0; j < DM; ++j) J[i][j] += x[i] * dn[j];
        }
        det_inv<DM>(J, inv);
        double flux[DM];
        if (dir_or_null) {
            for (int d = 0; d < DM; ++d) flux[d] = dir_or_null[d];
        } else {
            double nrm = 0.0;
            for (int j = 0; j < DM; ++j) {
                double v = 0.0;
                for (int i = 0; i < DM; ++i) v += ft_normal[tip * DM + i] * inv[i][j];
                flux[j] = v;
                nrm += v * v;
            }
            nrm = sqrt(nrm) + 1.e-30;
            for (int d = 0; d < DM; ++d) flux[d] /= nrm;
        }
        const double axw = size * ft_weight[tip];
        for (int d = 0; d < DM; ++d) flux[d] = traction * flux[d] * axw;
        for (int32_t fn = 0; fn < nfn; ++fn) {
            const double shape = ft_N[tip * npe + key[fn]];
            for (int d = 0; d < DM; ++d) contrib[(int64_t)fn * DM + d] += flux[d] * shape;
        }
    }
}

}  // namespace femcy
