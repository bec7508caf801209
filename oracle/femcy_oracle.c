/*
 * TEST INFRASTRUCTURE ONLY -- C restatement of FEMcy's hot kernels AS WRITTEN (storage layouts,
 * loop structure, per-entry linear search, atomic adds, 8 vector passes + 4 reductions per CG
 * iteration), parallelised with OpenMP the way Taichi's ti.cpu backend parallelises a struct-for.
 * Used (a) as a second, independently written checker next to oracle/femcy_oracle.py and (b) as
 * the timed CPU baseline of bench.py ("cpu_baseline.kind = port": the real Taichi run cannot be
 * produced here).  Nothing in femcy_amd/ links or loads this file.
 *
 * PARITY STATUS: "parity unpinned" against a real Taichi run; pinned through femcy_oracle.py
 * (tests/test_oracle_c.py cross-checks every function here against it) and -- round 5 -- by the reference's
 * published sigma_yy values: with serial sums this CG stops where FEMcy's README numbers are (93.56 after 105
 * iterations on the CPS3 deck, 93.32 / 84.40 after 128 on the CPS6 deck;
 * test_as_written_cg_reproduces_all_three_published_numbers).
 *
 * file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXNPE 10
#define MAXDM 3
#define MAXM (MAXNPE * MAXDM)
#define MAXS 6

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* tests pin the reduction order: 1 thread = the sums of a serial loop, as written */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* closed-form inverse / determinant of ti.Matrix for 2x2 and 3x3 */
static double inv_det(int dm, const double* J, double* inv) {
    if (dm == 2) {
        double det = J[0] * J[3] - J[1] * J[2];
        inv[0] = J[3] / det; inv[1] = -J[1] / det; inv[2] = -J[2] / det; inv[3] = J[0] / det;
        return det;
    }
    double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    double det = J[0] * c00 + J[1] * c01 + J[2] * c02;
    inv[0] = c00 / det; inv[3] = c01 / det; inv[6] = c02 / det;
    inv[1] = (J[2] * J[7] - J[1] * J[8]) / det; inv[4] = (J[0] * J[8] - J[2] * J[6]) / det; inv[7] = (J[1] * J[6] - J[0] * J[7]) / det;
    inv[2] = (J[1] * J[5] - J[2] * J[4]) / det; inv[5] = (J[2] * J[3] - J[0] * J[5]) / det; inv[8] = (J[0] * J[4] - J[1] * J[3]) / det;
    return det;
}

/* ddsdde_init (stiffnessMtrx.py:124-129): one copy of C per Gauss point, as the reference stores it */
void orc_ddsdde_init(int64_t ngp_total, int s, const double* C, double* ddsdde) {
#pragma omp parallel for
    for (int64_t k = 0; k < ngp_total; ++k) memcpy(ddsdde + k * s * s, C, sizeof(double) * s * s);
}

/* get_dsdx_and_vol (stiffnessMtrx.py:132-150); use_dof = 0 gives dsdX at the reference configuration */
void orc_get_dsdx_and_vol(int ne, int npe, int dm, int nGP, const double* nodes, const double* dof, int use_dof,
                          const int32_t* elements, const double* dN, const double* w, double* dsdx, double* vol) {
#pragma omp parallel for
    for (int ele = 0; ele < ne; ++ele) {
        double x[MAXNPE][MAXDM];
        for (int i = 0; i < npe; ++i)
            for (int j = 0; j < dm; ++j) {
                int nd = elements[(int64_t)ele * npe + i];
                x[i][j] = nodes[(int64_t)nd * dm + j] + (use_dof ? dof[(int64_t)nd * dm + j] : 0.0);
            }
        for (int g = 0; g < nGP; ++g) {
            const double* dsdn = dN + (int64_t)g * npe * dm;
            double J[9], inv[9];
            for (int i = 0; i < dm; ++i)
                for (int j = 0; j < dm; ++j) {
                    double a = 0.0;
                    for (int k = 0; k < npe; ++k) a += x[k][i] * dsdn[k * dm + j];
                    J[i * dm + j] = a;
                }
            double det = inv_det(dm, J, inv);
            double* out = dsdx + ((int64_t)ele * nGP + g) * npe * dm;
            for (int a = 0; a < npe; ++a)
                for (int j = 0; j < dm; ++j) {
                    double acc = 0.0;
                    for (int k = 0; k < dm; ++k) acc += dsdn[a * dm + k] * inv[k * dm + j];
                    out[a * dm + j] = acc;
                }
            vol[(int64_t)ele * nGP + g] = det * w[g];
        }
    }
}

/* strainMtrx (every element_zoo class): B[s][m] from dsdx[npe][dm] */
static void strain_mtrx(int npe, int dm, const double* g, double* B, int m) {
    int s = dm == 2 ? 3 : 6;
    memset(B, 0, sizeof(double) * s * m);
    for (int a = 0; a < npe; ++a) {
        if (dm == 2) {
            B[0 * m + 2 * a] = g[a * 2];     B[1 * m + 2 * a + 1] = g[a * 2 + 1];
            B[2 * m + 2 * a] = g[a * 2 + 1]; B[2 * m + 2 * a + 1] = g[a * 2];
        } else {
            B[0 * m + 3 * a] = g[a * 3];         B[1 * m + 3 * a + 1] = g[a * 3 + 1]; B[2 * m + 3 * a + 2] = g[a * 3 + 2];
            B[3 * m + 3 * a] = g[a * 3 + 1];     B[3 * m + 3 * a + 1] = g[a * 3];
            B[4 * m + 3 * a] = g[a * 3 + 2];     B[4 * m + 3 * a + 2] = g[a * 3];
            B[5 * m + 3 * a + 1] = g[a * 3 + 2]; B[5 * m + 3 * a + 2] = g[a * 3 + 1];
        }
    }
}

/* sparseMatrix_get_j (stiffnessMtrx.py:414-420): linear search of the column slot */
static inline int get_j(const int32_t* ij, int W1, int64_t i_global, int32_t j_global) {
    const int32_t* row = ij + i_global * W1;
    int j_local = 0;
    for (int j = 0; j < row[0]; ++j)
        if (row[j + 1] == j_global) j_local = j;
    return j_local;
}

/* assemble_stiffnessMtrx (stiffnessMtrx.py:161-186): zero-fill, then per (element, Gauss point)
 * dense B^T (C B), per-entry search + atomic add into sparseMtrx_rowMajor f64[n][W] */
void orc_assemble(int ne, int npe, int dm, int nGP, const int32_t* elements, const double* dsdx, const double* vol,
                  const double* ddsdde, const int32_t* ij, int W, int64_t n, double* A) {
    const int m = npe * dm, s = dm == 2 ? 3 : 6, W1 = W + 1;
#pragma omp parallel for
    for (int64_t k = 0; k < n * W; ++k) A[k] = 0.0;
#pragma omp parallel for
    for (int64_t eg = 0; eg < (int64_t)ne * nGP; ++eg) {
        const int64_t ele = eg / nGP;
        double B[MAXS * MAXM], CB[MAXS * MAXM], bcb[MAXM * MAXM];
        strain_mtrx(npe, dm, dsdx + eg * npe * dm, B, m);
        const double* C = ddsdde + eg * s * s;
        for (int p = 0; p < s; ++p)
            for (int c = 0; c < m; ++c) {
                double a = 0.0;
                for (int q = 0; q < s; ++q) a += C[p * s + q] * B[q * m + c];
                CB[p * m + c] = a;
            }
        for (int r = 0; r < m; ++r)
            for (int c = 0; c < m; ++c) {
                double a = 0.0;
                for (int p = 0; p < s; ++p) a += B[p * m + r] * CB[p * m + c];
                bcb[r * m + c] = a;
            }
        int32_t Js[MAXM];
        for (int a = 0; a < npe; ++a)
            for (int i = 0; i < dm; ++i) Js[a * dm + i] = elements[ele * npe + a] * dm + i;
        const double v = vol[eg];
        for (int node = 0; node < npe; ++node)
            for (int i_local = 0; i_local < dm; ++i_local) {
                const int64_t i_global = (int64_t)elements[ele * npe + node] * dm + i_local;
                for (int j_local = 0; j_local < m; ++j_local) {
                    const int j = get_j(ij, W1, i_global, Js[j_local]);
                    const double add = bcb[(node * dm + i_local) * m + j_local] * v;
#pragma omp atomic
                    A[i_global * W + j] += add;
                }
            }
    }
}

/* get_deformation_gradient (stiffnessMtrx.py:532-556) */
void orc_deformation_gradient(int ne, int npe, int dm, int nGP, const double* nodes, const double* dof,
                              const int32_t* elements, const double* dN, double* F) {
#pragma omp parallel for
    for (int ele = 0; ele < ne; ++ele) {
        double X[MAXNPE][MAXDM], U[MAXNPE][MAXDM];
        for (int a = 0; a < npe; ++a)
            for (int j = 0; j < dm; ++j) {
                int nd = elements[(int64_t)ele * npe + a];
                U[a][j] = dof[(int64_t)nd * dm + j];
                X[a][j] = nodes[(int64_t)nd * dm + j];
            }
        for (int g = 0; g < nGP; ++g) {
            const double* dsdn = dN + (int64_t)g * npe * dm;
            double J[9], inv[9];
            for (int i = 0; i < dm; ++i)
                for (int j = 0; j < dm; ++j) {
                    double a = 0.0;
                    for (int k = 0; k < npe; ++k) a += X[k][i] * dsdn[k * dm + j];
                    J[i * dm + j] = a;
                }
            inv_det(dm, J, inv);
            double* Fo = F + ((int64_t)ele * nGP + g) * dm * dm;
            for (int i = 0; i < dm; ++i)
                for (int j = 0; j < dm; ++j) {
                    double acc = 0.0;
                    for (int a = 0; a < npe; ++a) {
                        double dsdX = 0.0;
                        for (int k = 0; k < dm; ++k) dsdX += dsdn[a * dm + k] * inv[k * dm + j];
                        acc += U[a][i] * dsdX;
                    }
                    Fo[i * dm + j] = acc + (i == j ? 1.0 : 0.0);
                }
        }
    }
}

static double det3(const double* A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* sigma = F S F^T / det F with S from Voigt [xx,yy,zz,xy,zx,yz] */
static void stvk3(const double* F, const double* C6, double* sig) {
    double E[9], ev[6], sv[6], S[9], FS[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += F[k * 3 + i] * F[k * 3 + j];
            E[i * 3 + j] = (a - (i == j ? 1.0 : 0.0)) / 2.0;
        }
    ev[0] = E[0]; ev[1] = E[4]; ev[2] = E[8]; ev[3] = 2.0 * E[1]; ev[4] = 2.0 * E[6]; ev[5] = 2.0 * E[5];
    for (int p = 0; p < 6; ++p) {
        double a = 0.0;
        for (int q = 0; q < 6; ++q) a += C6[p * 6 + q] * ev[q];
        sv[p] = a;
    }
    S[0] = sv[0]; S[4] = sv[1]; S[8] = sv[2]; S[1] = S[3] = sv[3]; S[2] = S[6] = sv[4]; S[5] = S[7] = sv[5];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) FS[i * 3 + j] = F[i * 3] * S[j] + F[i * 3 + 1] * S[3 + j] + F[i * 3 + 2] * S[6 + j];
    double J = det3(F);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            sig[i * 3 + j] = (FS[i * 3] * F[j * 3] + FS[i * 3 + 1] * F[j * 3 + 1] + FS[i * 3 + 2] * F[j * 3 + 2]) / J;
}

/* constitutiveOfLargeDeform: kind 0 lin3d (linear_isotropic.py:55-76), 1 plane strain
 * (linear_isotropic_plane_strain.py:66-86), 2 plane stress (linear_isotropic_plane_stress.py:65-96,
 * uses C_6x6 built from E, nu), 3 neo-Hookean (neo_hookean.py:66-77) */
void orc_cauchy_large(int64_t ngp_total, int dm, int kind, const double* ddsdde, double p0, double p1, const double* F,
                      double* sigma) {
#pragma omp parallel for
    for (int64_t k = 0; k < ngp_total; ++k) {
        const double* Fk = F + k * dm * dm;
        double* sk = sigma + k * dm * dm;
        if (kind == 0) {
            stvk3(Fk, ddsdde + k * 36, sk);
        } else if (kind == 3) {
            double J = det3(Fk);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double B = Fk[i * 3] * Fk[j * 3] + Fk[i * 3 + 1] * Fk[j * 3 + 1] + Fk[i * 3 + 2] * Fk[j * 3 + 2];
                    double eye = i == j ? 1.0 : 0.0;
                    sk[i * 3 + j] = 2.0 * p0 / J * (B - eye) + 2.0 * p1 * (J - 1.0) * eye;
                }
        } else if (kind == 1) {
            const double* C = ddsdde + k * 9;
            double E[4];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    E[i * 2 + j] = (Fk[i] * Fk[j] + Fk[2 + i] * Fk[2 + j] - (i == j ? 1.0 : 0.0)) / 2.0;
            double ev[3] = {E[0], E[3], E[1] + E[2]}, sv[3];
            for (int p = 0; p < 3; ++p) sv[p] = C[p * 3] * ev[0] + C[p * 3 + 1] * ev[1] + C[p * 3 + 2] * ev[2];
            double S[4] = {sv[0], sv[2], sv[2], sv[1]}, FS[4];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) FS[i * 2 + j] = Fk[i * 2] * S[j] + Fk[i * 2 + 1] * S[2 + j];
            double J = Fk[0] * Fk[3] - Fk[1] * Fk[2];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) sk[i * 2 + j] = (FS[i * 2] * Fk[j * 2] + FS[i * 2 + 1] * Fk[j * 2 + 1]) / J;
        } else {
            const double E_ = p0, nu = p1, G = E_ / 2.0 / (1.0 + nu), c00 = E_ / (1.0 - nu * nu), c01 = c00 * nu;
            double C6[36] = {0};
            C6[0] = C6[7] = c00; C6[1] = C6[6] = c01; C6[21] = G;
            double F3[9] = {Fk[0], Fk[1], 0, Fk[2], Fk[3], 0, 0, 0, 0}, s3[9];
            F3[8] = -nu / (1.0 - nu) * (Fk[0] + Fk[3] - 2.0) + 1.0;
            stvk3(F3, C6, s3);
            sk[0] = s3[0]; sk[1] = s3[1]; sk[2] = s3[3]; sk[3] = s3[4];
        }
    }
}

/* assemble_nodal_force_GN_kernel (stiffnessMtrx.py:620-644): node-parallel gather over the padded
 * nodeEles table (-1 = empty), local index found by search (tiGadgets.py:94-101) */
void orc_nodal_force(int nn, int npe, int dm, int nGP, int maxEles, const int32_t* nodeEles, const int32_t* elements,
                     const double* dsdx, const double* sigma, const double* vol, double* f) {
#pragma omp parallel for
    for (int node0 = 0; node0 < nn; ++node0) {
        for (int i = 0; i < dm; ++i) f[(int64_t)node0 * dm + i] = 0.0;
        for (int ie = 0; ie < maxEles; ++ie) {
            int ele = nodeEles[(int64_t)node0 * maxEles + ie];
            if (ele == -1) continue;
            int nid = -1;
            for (int a = 0; a < npe; ++a)
                if (elements[(int64_t)ele * npe + a] == node0) nid = a;
            for (int g = 0; g < nGP; ++g) {
                const double* gr = dsdx + (((int64_t)ele * nGP + g) * npe + nid) * dm;
                const double* sg = sigma + ((int64_t)ele * nGP + g) * dm * dm;
                for (int i = 0; i < dm; ++i) {
                    double d = 0.0;
                    for (int j = 0; j < dm; ++j) d += gr[j] * sg[j * dm + i];
                    f[(int64_t)node0 * dm + i] = f[(int64_t)node0 * dm + i] + d * vol[(int64_t)ele * nGP + g];
                }
            }
        }
    }
}

/* ----------------------------------------------------------------------------------------- CG
 * ConjugateGradientSolver_rowMajor (conjugateGradientSolver.py:10-127): every kernel is its own pass. */
static double A_get(const double* A, const int32_t* ij, int W, int64_t i, int32_t j) {   /* :40-46 */
    int target = 0;
    const int32_t* row = ij + i * (W + 1);
    for (int j0 = 0; j0 < row[0]; ++j0)
        if (row[j0 + 1] == j) target = j0;
    return A[i * W + target];
}
void orc_compute_Ad(int64_t n, int W, const double* A, const int32_t* ij, const double* d, double* Ad) {   /* :53-58 */
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* row = ij + i * (W + 1);
        double acc = 0.0;
        for (int j0 = 0; j0 < row[0]; ++j0) acc = acc + A[i * W + j0] * d[row[j0 + 1]];
        Ad[i] = acc;
    }
}
static double k_rmax(int64_t n, const double* r) {
    double rm = 0.0;
#pragma omp parallel for reduction(max : rm)
    for (int64_t i = 0; i < n; ++i) rm = fmax(rm, fabs(r[i]));
    return rm;
}
static double k_rMr(int64_t n, const double* r, const double* M) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) s += r[i] * M[i] * r[i];
    return s;
}
static double k_dot(int64_t n, const double* y, const double* z) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) s += y[i] * z[i];
    return s;
}

/* returns the number of loop bodies executed; work = 5*n doubles (x r d M Ad are caller-provided) */
int orc_cg_solve(int64_t n, int W, const double* A, const int32_t* ij, const double* b, double eps, int maxit, double* x,
                 double* r, double* d, double* M, double* Ad, double* r0_out, double* rmax_out) {
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {   /* re_init + M_init */
        x[i] = 0.0;
        Ad[i] = 0.0;
        M[i] = 1.0 / A_get(A, ij, W, i, (int32_t)i);
    }
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) r[i] = b[i];          /* r_d_init */
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) d[i] = M[i] * r[i];
    const double r0 = k_rmax(n, r);
    double rmax = r0;
    int it = 0;
    const int64_t lim = maxit > 0 ? maxit : n;
    for (int64_t i = 0; i < lim; ++i) {
        orc_compute_Ad(n, W, A, ij, d, Ad);
        const double rMr = k_rMr(n, r, M);
        const double alpha = rMr / k_dot(n, d, Ad);
#pragma omp parallel for
        for (int64_t j = 0; j < n; ++j) x[j] = x[j] + alpha * d[j];
#pragma omp parallel for
        for (int64_t j = 0; j < n; ++j) r[j] = r[j] - alpha * Ad[j];
        const double beta = k_rMr(n, r, M) / rMr;
#pragma omp parallel for
        for (int64_t j = 0; j < n; ++j) d[j] = M[j] * r[j] + beta * d[j];
        rmax = k_rmax(n, r);
        it = (int)(i + 1);
        if (rmax < eps * r0) break;
    }
    if (r0_out) *r0_out = r0;
    if (rmax_out) *rmax_out = rmax;
    return it;
}

/* sparseIJ builder for big meshes (stiffnessMtrx.py:78-89 restated from CSR adjacency; the reference's
 * Python loops would take minutes at 1M elements and are not part of the timed path) */
void orc_build_sparseIJ(int nn, int dm, const int64_t* adj_ptr, const int64_t* adj_idx, int W, int32_t* ij) {
#pragma omp parallel for
    for (int node0 = 0; node0 < nn; ++node0) {
        int len = (int)(adj_ptr[node0 + 1] - adj_ptr[node0]) * dm;
        for (int i = 0; i < dm; ++i) {
            int32_t* row = ij + ((int64_t)node0 * dm + i) * (W + 1);
            row[0] = len;
            int w = 1;
            for (int64_t k = adj_ptr[node0]; k < adj_ptr[node0 + 1]; ++k)
                for (int c = 0; c < dm; ++c) row[w++] = (int32_t)(adj_idx[k] * dm + c);
            for (; w <= W; ++w) row[w] = -1;
        }
    }
}
