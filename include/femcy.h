/*
 * femcy.h -- C ABI of libfemcy_hip.so, the MI355X (gfx950) solve path behind FEMcy's Python surface.
 *
 * The reference (mo-hanxuan/FEMcy) has no FFI: its "operator API" is Python duck typing over Taichi
 * kernels.  Every entry point below names the reference kernel(s)/method(s) it replaces
 * (file:line relative to the reference checkout) -- this is exactly the set a ctypes binding of
 * `System_of_equations` / `ConjugateGradientSolver_rowMajor` needs (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, a negative FEMCY_E* code otherwise; the message is
 *     available from femcy_last_error() (thread-local).  The library never aborts or throws.
 *   - host pointers are caller-owned and only borrowed for the duration of the call.
 *   - device memory is library-owned inside the opaque femcy_ctx: one ctx = one HIP device + one
 *     stream.  A ctx is not thread-safe; different ctxs are independent.
 *   - all floating point is f64, all indices i32 (reference: main.py:11 default_fp=ti.f64).
 *   - DOF numbering: i = node*dm + component (stiffnessMtrx.py:179-180).
 *   - solver vectors (the reference's ti.fields rhs/dof/residual/...) live in HBM inside the ctx and
 *     are addressed by the femcy_vec ids below; femcy_vec_upload/_download move them across PCIe.
 */
#ifndef FEMCY_H
#define FEMCY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct femcy_ctx femcy_ctx;

enum femcy_status {
    FEMCY_OK = 0,
    FEMCY_EINVAL = -1,     /* bad argument / call order */
    FEMCY_EHIP = -2,       /* a HIP runtime call failed */
    FEMCY_ENOKERNEL = -3,  /* no kernel instantiation for this (npe, dm, nGP) / material */
    FEMCY_ENUMERIC = -4,   /* NaN/breakdown detected by a solver */
    FEMCY_ECOMM = -5,      /* RCCL failure / a rendezvous of the in-process group that did not complete */
    FEMCY_ENOMEM = -6
};

/* device-resident solver vectors, each of length n = nn*dm (stiffnessMtrx.py:33-34,55-56,95,113;
 * conjugateGradientSolver.py:20-29) */
enum femcy_vec {
    FEMCY_VEC_DOF = 0,       /* System_of_equations.dof                 */
    FEMCY_VEC_RHS = 1,       /* .rhs                                    */
    FEMCY_VEC_RESIDUAL = 2,  /* .residual_nodal_force                   */
    FEMCY_VEC_FORCE = 3,     /* .nodal_force                            */
    FEMCY_VEC_DU = 4,        /* .du                                     */
    FEMCY_VEC_DOF_OLD = 5,   /* .dof_old                                */
    FEMCY_VEC_X = 6,         /* ConjugateGradientSolver_rowMajor.x      */
    FEMCY_VEC_TMP0 = 7,
    FEMCY_VEC_TMP1 = 8,
    FEMCY_VEC_COUNT = 9
};

/* Voigt pattern of the strain matrix B (element_zoo: strainMtrx) */
enum femcy_voigt { FEMCY_VOIGT_2D = 0 /* xx,yy,xy */, FEMCY_VOIGT_3D = 1 /* xx,yy,zz,xy,zx,yz */ };

/* material_zoo classes (reader/inp_info.py:294-316) */
enum femcy_material {
    FEMCY_MAT_LIN3D = 0,    /* linear_isotropic.py            params = {E, nu}  */
    FEMCY_MAT_PSTRAIN = 1,  /* linear_isotropic_plane_strain  params = {E, nu}  */
    FEMCY_MAT_PSTRESS = 2,  /* linear_isotropic_plane_stress  params = {E, nu}  */
    FEMCY_MAT_NEOHOOKE = 3  /* neo_hookean.py                 params = {C1, D1}; with dm = 2: plane strain
                               (extension, the reference has the 3-D form only)                          */
};

/* Gauss-point fields that can be downloaded for checking (stiffnessMtrx.py:40-61) */
enum femcy_gpfield {
    FEMCY_GP_DSDX = 0,   /* f64[ne][nGP][npe][dm] */
    FEMCY_GP_VOL = 1,    /* f64[ne][nGP]          */
    FEMCY_GP_F = 2,      /* f64[ne][nGP][dm][dm]  */
    FEMCY_GP_SIGMA = 3,  /* f64[ne][nGP][dm][dm]  */
    FEMCY_GP_STRAIN = 4, /* f64[ne][nGP][dm][dm]  (after femcy_compute_strain_stress) */
    FEMCY_GP_MISES = 5,  /* f64[ne][nGP]          */
    FEMCY_GP_ENERGY = 6  /* f64[ne][nGP] elastic energy density (after femcy_elastic_energy) */
};

/* assembly strategy (femcy_set_option FEMCY_OPT_ASSEMBLY) */
enum femcy_assembly {
    FEMCY_ASM_GATHER = 0, /* owner-computes: one lane per stored block, deterministic            */
    FEMCY_ASM_ATOMIC = 1, /* element scatter with f64 HW atomics (comparison / race check)        */
    FEMCY_ASM_ROWS = 2,   /* one wavefront per matrix row, LDS reduction, deterministic           */
    FEMCY_ASM_AUTO = 3,   /* default: ROWS4 for C3D10 (ROWS2 / ROWS if its LDS does not fit), PAIRS for the 2-D quadratic
                             families (ROWS if its LDS does not fit), GATHER_SYM(_ROWSUM) otherwise */
    FEMCY_ASM_GATHER_SYM = 4, /* GATHER on the diagonal + upper blocks only, mirrored stores of K_ba = K_ab^T: 1.3x on C3D4 */
    FEMCY_ASM_GATHER_SYM_ROWSUM = 5, /* the same with the diagonal block from K_aa = -sum_{b != a} K_ab (partition of
                                unity, checked on the element tables); AUTO picks it for npe <= 4 */
    FEMCY_ASM_ROWS2 = 6,  /* ROWS with the element records staged in LDS, one workgroup per 64-row slice, row-sum
                             diagonal, geometric-sum blocks for cubic-pattern C: AUTO's choice for C3D10 until round 3;
                             instantiated for C3D10 and C3D4 tables whose gradients sum to zero */
    FEMCY_ASM_ROWS3 = 7,  /* ROWS2 with eight adjacent rows finished together and written as whole 128-byte lines
                             (no read-for-fill of K: round 3) */
    FEMCY_ASM_ROWS4 = 8,  /* two rows per wavefront at a time (half a wave each, three incident elements per step, the
                             diagonal block computed like the others): round 3, C3D10; AUTO picks it there */
    FEMCY_ASM_PAIRS = 9   /* round 6: a wavefront owns 8 (or 16) adjacent rows of a slice; lanes = (row, incident element) pair x
                             column node, the element's whole record is read once per pair by coalesced 16-byte loads
                             (the row node's gradients come from the neighbouring lane), the geometric sums are reduced
                             in a wave-private LDS tile and the tile is written as 256-byte runs; any constant C.
                             2-D families; AUTO picks it for CPE6 / CPS6 / CPE8 / CPS8 */
};

enum femcy_option {
    FEMCY_OPT_ASSEMBLY = 0,     /* enum femcy_assembly, default AUTO                           */
    FEMCY_OPT_PCG_POLL = 1,     /* iterations between host polls of the device "done" flag      */
    FEMCY_OPT_TIMING = 2,       /* 1 = time kernel classes with hipEvents (femcy_timing);       */
                                /* k > 1 = same, but only every k-th SpMV launch is sampled     */
    FEMCY_OPT_SPMV_VARIANT = 3, /* SpMV wavefronts per 64-node slice: 0 auto (by mean row length), 1, 2, 4 */
    FEMCY_OPT_EW_GRID = 4,      /* cap on workgroups of the element-wise PCG kernels (tuning)   */
    FEMCY_OPT_SELL_SIGMA = 6,   /* rows are sorted by length inside windows of this many nodes before slicing
                                   (SELL-C-sigma, default 4096; 64 = natural order); set before build_pattern */
    FEMCY_OPT_PCG_GRAPH = 5,    /* hipGraph replay of poll-bursts of PCG iterations: 0 off, 1 auto (default:
                                   below 2e5 DOF, where the loop is launch-bound), 2 always             */
    FEMCY_OPT_EXCHANGE = 8,     /* multi-rank interface exchange: 0 = all-reduce of the packed global interface vector
                                   (default), 1 = send/recv with the neighbouring ranks (needs
                                   femcy_comm_set_neighbours); femcy_comm_tune measures both and sets it */
    FEMCY_OPT_PCG_STORAGE_ORDER = 13, /* 1 (default): the three-launch PCG of a single rank keeps its vectors in the matrix's
                                   storage order (b permuted once, x permuted back at the end): lanes of a wavefront are
                                   rows of one length class in ascending order, so their gathers of d touch neighbouring
                                   addresses; 0 = node order (what multi-rank runs keep)                          */
    FEMCY_OPT_PCG_FUSED_UPDATE = 15, /* 0 (default) = two vector kernels per iteration of the three-launch PCG; 1 = ONE (r update,
                                   the grid-wide (r.M.r, max|r|) as an in-kernel exchange of tagged granules, d and x
                                   update; single rank, <= 8 double2 per thread at <= 1024 resident workgroups; a
                                   time-out of the exchange falls back to the two kernels for good).  Built and measured
                                   in round 4: the exchange (2.6-3.3 us) costs more than the kernel boundary it removes
                                   (1.5 us) -- 80.5 -> 82.0 us per iteration on the C3D10 plate, 42.3 -> 43.8 at 1 M C3D4
                                   (profiles/r04_ab_fused_update.txt) -- so it stays an option                    */
    FEMCY_OPT_SPMV_FOOTPRINT = 16, /* storage-order product of the three-launch PCG: 1 = every wave first stages the x entries
                                   of its footprint (the sorted distinct positions its block rows refer to) in LDS with
                                   coalesced loads and gathers from there through 16-bit local columns; 0 = gathers from
                                   global memory per block (default: measured 0 ... 6 % faster than the footprint form,
                                   profiles/r04_ab_footprint_product.txt)                                          */
    FEMCY_OPT_DIRECT_MAX_BYTES = 17, /* femcy_direct_solve: largest band (bytes) it may allocate (default 48 GiB); a system
                                   whose band after reverse Cuthill-McKee is larger is refused with FEMCY_ENOMEM    */
    FEMCY_OPT_NODE_ORDER = 14,  /* internal row order of the matrix, set before femcy_build_pattern; vectors handed
                                   to / from the caller always keep the caller's numbering.  0 = rows sorted by length
                                   inside windows of the caller's numbering; 1 (default) = inside windows of the best of the
                                   lexicographic coordinate orders (one per axis permutation) if its measured gather cost
                                   (cache lines per wavefront gather, femcy_get_node_order) beats the caller's numbering
                                   by 10 % -- otherwise the caller's numbering; 2 + k = coordinate order k forced (tests).
                                   Measured: persistent PCG on C3D10 plates of 37 k / 72 k elements 31.8 -> 28.8 / 40.7 ->
                                   34.9 us per iteration; neutral for the three-launch PCG and the assemblies; the 1 M / 8 M
                                   C3D4 plates keep their numbering (profiles/r04_persist_node_order_c3d10.txt)      */
    FEMCY_OPT_PCG_PERSIST = 11, /* 1 (default): single-rank systems that fit one wavefront task per SIMD (3 x 3 blocks: up
                                   to ~7.8e5 DOF on MI355X; 2 x 2 blocks: up to ~1.05e6) and fill the chip 1.5 times over
                                   are solved by one persistent launch -- vectors and part of the matrix in registers,
                                   another part in LDS, the rest streamed (from the Infinity Cache or, since round 5, from
                                   HBM: FEMCY_TUNE_PERSIST_MAX_MB), grid-wide exchanges at the three synchronisation points
                                   of the recurrence; 2 = any system whose slices fit (tests); 0 = three launches per
                                   iteration */
    FEMCY_OPT_PCG_PERSIST_MULTI = 12, /* 1 (default): with a communicator attached, femcy_pcg keeps the one-launch
                                   persistent kernel on every rank and the ranks' kernels exchange the interface rows of
                                   Ad and the two scalar reductions through mailboxes in each other's HBM (written over
                                   xGMI by the peers' kernels) -- once the mailboxes are exchanged
                                   (femcy_comm_mailbox_export / _import) and EVERY rank agreed
                                   (femcy_comm_persist_agree).  0 = three launches + RCCL calls per iteration */
    FEMCY_OPT_PCG_SMALL = 10,   /* 1 (default): systems whose two work vectors fit the LDS of a workgroup (~1e4 DOF on
                                   MI355X) are solved by ONE persistent launch with one grid barrier per iteration
                                   instead of three launches per iteration; 0 = always the three-kernel loop */
    FEMCY_OPT_OVERLAP = 9,      /* multi-rank PCG with the neighbour exchange: 1 (default) = the slices that hold interface
                                   nodes are multiplied first and their exchange runs on a second stream while the
                                   interior slices are multiplied; 0 = everything on one stream */
    FEMCY_OPT_TANGENT = 7,      /* what femcy_assemble_K assembles.  0 (default) = the reference's matrix: B^T C B
                                   on the current configuration with the constant C (stiffnessMtrx.py:124-186).
                                   1 = the consistent tangent of femcy_internal_force: spatial elasticity tensor of
                                   the material at F (the updates the reference left commented out,
                                   neo_hookean.py:62-64, 79-81) + geometric stiffness (grad N_a . sigma . grad N_b) I.
                                   An EXTENSION outside the parity runs: Newton iterates differ from the
                                   reference's (they converge quadratically).  Isotropic 3-D, plane-strain and
                                   neo-Hookean materials; not available for plane stress. */

    /* ---- tuning and test knobs.  Not needed by a caller of the path; the tests and the tools under tools/ use them
     * to force code paths and to measure alternatives.  None of them changes results beyond summation order. */
    FEMCY_TUNE_TIMING_FENCE = 100,   /* 1 (default): an empty kernel precedes every SpMV dispatch that is timed with
                                        dispatch-attached events, so that the start stamp is not taken while the
                                        previous kernel drains                                                     */
    FEMCY_TUNE_SPMV_WG_PER_XCD = 101,/* SpMV workgroups per XCD; longer slice ranges are looped inside the kernel.  0 (default)
                                        = 512 where an XCD's range holds more than 512 tasks of a matrix beyond 512 MiB, else
                                        256 (round 6: 5 workgroups per CU are resident, the rest is handed out as CUs become
                                        free; 3 GB C3D10 product 597 -> 560 us, profiles/r06_spmv_rounds.txt); 1 forces the
                                        loop on small meshes (tests)                                               */
    FEMCY_TUNE_SPMV_NT = 102,        /* SpMV matrix stream non-temporal: -1 auto (stored matrix > 256 MiB), 0, 1    */
    FEMCY_TUNE_VEC_NT = 103,         /* PCG vector kernels non-temporal: -1 auto (vector > 12 MB), 0, 1             */
    FEMCY_TUNE_PERSIST_LDS_ROWS = 104,/* persistent PCG: block rows per wave kept in LDS (-1 = as many as fit)      */
    FEMCY_TUNE_PERSIST_REG_ROWS = 105,/* persistent PCG: block rows per slice kept in registers (0, 4 or 5)         */
    FEMCY_TUNE_PERSIST_PROBE = 106,  /* persistent PCG: 16 = no prefetch during the barriers.  Bits 0-3 (skip the
                                        streamed / LDS / register rows, skip the barrier wait: timing experiments
                                        that produce meaningless numbers) are accepted only by a library built with
                                        -DFEMCY_PERSIST_PROBE; the shipped library has no work-skipping path       */
    FEMCY_TUNE_PERSIST_WGS = 107,    /* persistent PCG: workgroups of the launch (0 = one per CU).  More than the
                                        occupancy query admits is refused before the launch; see femcy_pcg          */
    FEMCY_TUNE_SMALL_REG_ROWS = 108, /* small-system PCG: block rows per wave kept in registers (-1 = its share)    */
    FEMCY_TUNE_PERSIST_VARIANT = 109,/* persistent PCG variant bits (-1 = the default chosen by measurement, DESIGN.md
                                        section 3): 1 = non-temporal matrix stream (FEMCY_TUNE_PERSIST_L2_ROWS), 2 = the
                                        three grid-wide exchanges as tagged granules (one hop) instead of counters +
                                        data, 4 = d published in storage order (16 + 8 byte gathers instead of 3 x 8).
                                        The shipped library holds the default and 0 (round 2); all eight with
                                        -DFEMCY_PERSIST_ALL_VARIANTS */
    FEMCY_TUNE_SPMV_KEEP = 110,      /* NT SpMV: per-mille of every XCD's slice range that keeps the default cache
                                        policy (-1 auto = 235 MB of the matrix)                                     */
    FEMCY_TUNE_SKIP_OCCUPANCY_CHECK = 111,/* 1 = launch the persistent kernels without the co-residency check (tests
                                        of the barrier time-out and its fallback)                                   */
    FEMCY_TUNE_PERSIST_L2_ROWS = 113,/* persistent PCG with non-temporal matrix stream (variant bit 1): streamed block rows
                                        per slice that keep the default cache policy (they stay in the XCD's L2)     */
    FEMCY_TUNE_DIRECT_UPDATE = 115,  /* femcy_direct_solve, trailing tile update: -1 auto (default), 0 = VALU product, 1 = f64
                                        matrix cores (v_mfma_f64_16x16x4_f64), one tile pair per workgroup, 2 = matrix
                                        cores, 2 x 2 tile pairs per workgroup, 3 = 1 on two streams: the tiles the next panel
                                        needs first, the rest beside that panel (tests, comparison records)          */
    FEMCY_TUNE_ROWS4_TILE = 116,     /* FEMCY_ASM_ROWS4 (C3D10), experiment of round 5: 1000 GP + LCUT = in slices no wider than
                                        LCUT blocks a wave owns 16 consecutive rows and writes 2 GP adjacent rows (GP 2 or 4)
                                        at a time from a tile of its own LDS (64 / 128 contiguous bytes per slot instead
                                        of 32); 0 = off (default: 288 - 295 us against 295, profiles/r05_pmc_rows4_tile.txt) */
    FEMCY_TUNE_ROWS4_ORDER = 118,    /* FEMCY_ASM_ROWS4 launch order: 0 = slices by decreasing work, round-robin over the XCDs (even
                                        shares of every weight class: best while the element records sit in the Infinity
                                        Cache), 1 = slices in Morton order of their centroids, XCD-contiguous ranges (records
                                        re-used inside one L2: best beyond it), -1 (default) = by the size of the records;
                                        the same bits of K either way */
    FEMCY_TUNE_SPMV_ROT = 119,       /* SpMV task lists of the workgroups of an XCD.  0 = task b + i * (workgroups per XCD) in round i
                                        (rounds 1-5: always the same position of the length-sorted window); 1 .. 63 = the
                                        position advances by this many tasks per round; 64 = lists balanced by the host, round
                                        by round (the workgroup with the most work so far takes the shortest task of the
                                        round); -1 (default) = 64 where the rows of a window differ in length by more than a
                                        quarter and an XCD's range holds more than 512 tasks of a matrix beyond 512 MiB (C3D10
                                        at k >= 8: 3 GB product 596 -> 529 us together with knob 101), 0 elsewhere.  Changes the grouping of the d.Ad partial sums only
                                        (profiles/r06_spmv_rounds.txt) */
    FEMCY_TUNE_PAIRS = 117,          /* FEMCY_ASM_PAIRS: -1 = default (163), else bit 0 = workgroups take XCD-contiguous ranges of
                                        the processing order, bits 1-2 = rows per wavefront (0: 16, 1: 8), bits 3-4 = steps of
                                        element records in flight - 2 (0..2), bit 5 = chunks processed in Morton order of their
                                        centroids, bits 6-9 = chunks per wavefront - 1; the same bits of K whatever the value
                                        (profiles/r06_asm_cpe8_knobs.txt) */
    FEMCY_TUNE_PERSIST_MAX_MB = 114, /* persistent PCG: largest STREAMED part of the matrix (MiB) it takes; 0 = no limit
                                        (default since round 5: 61 against 78 us per iteration on the 124 k C3D10 plate
                                        whose 287 MB stream comes from HBM); rounds 2-4 used 240 (tests, comparison
                                        records)                                                                     */
    FEMCY_TUNE_BARRIER_SPIN_LIMIT = 112 /* polls (each ~0.3-1 us) before a grid barrier of the one-launch solvers gives
                                        up, poisons the exchange and the solve is redone by the three-kernel loop
                                        (default 2^20, about half a second; 0 provokes the fallback: tests)         */
};

typedef struct femcy_pattern_info {
    int64_t n;           /* scalar DOFs                                                */
    int64_t nnzb;        /* structural dm x dm blocks (node adjacency incl. diagonal)   */
    int64_t nnz;         /* nnzb*dm*dm                                                  */
    int32_t max_row_blocks;   /* max neighbours per node (reference: maxLen, stiffnessMtrx.py:80) */
    int32_t ell_width;        /* reference W = max_row_blocks*dm                          */
    int64_t stored_blocks;    /* blocks stored incl. SELL padding                         */
    int32_t nslices;
    int32_t max_node_elems;   /* reference nodeEles width (stiffnessMtrx.py:71)           */
} femcy_pattern_info;

typedef struct femcy_timing_t {
    /* accumulated since the last femcy_timing_reset; *_ms from hipEvents on the ctx stream */
    double geom_ms;      int64_t geom_launches;      /* get_dsdx_and_vol / F / sigma kernels  */
    double assemble_ms;  int64_t assemble_launches;  /* K assembly kernel                     */
    double force_ms;     int64_t force_launches;     /* nodal-force gather                    */
    double spmv_ms;      int64_t spmv_launches;      /* compute_Ad                            */
    double pcg_ms;       int64_t pcg_iters;          /* whole PCG solves (all kernels)        */
    double persist_ms;   int64_t persist_launches;   /* one-launch PCG (k_pcg_persist) alone  */
    int64_t persist_iters;                           /* CG iterations inside those launches   */
    int64_t solves_three, solves_small, solves_persist; /* PCG solves by path (always counted) */
    int64_t barrier_timeouts;                        /* one-launch solves abandoned at a grid barrier and redone by
                                                        the three-kernel loop (always counted)              */
} femcy_timing_t;

/* ------------------------------------------------------------------ life cycle / diagnostics */
int femcy_ctx_create(int device, femcy_ctx** out);
int femcy_ctx_destroy(femcy_ctx* ctx);
const char* femcy_last_error(void);
int femcy_version(void);
int femcy_set_option(femcy_ctx* ctx, int option, int64_t value);
int femcy_sync(femcy_ctx* ctx);                                   /* hipStreamSynchronize */

/* ----------------------------------------------------------------------- problem definition */
/* Body + System_of_equations.__init__ state (body.py:13-17, stiffnessMtrx.py:26-121).  Calling it again on a used
 * ctx starts over: element tables, material, pattern, DOF lists and load sets of the old mesh are dropped. */
int femcy_set_mesh(femcy_ctx* ctx, int32_t nn, int32_t dm, const double* nodes /*[nn*dm]*/,
                   int32_t ne, int32_t npe, const int32_t* elems /*[ne*npe]*/);
/* table-driven element plugin: ELE.gaussPoints/gaussWeights/dshape_dnat (element_zoo modules) */
int femcy_set_element(femcy_ctx* ctx, int32_t nGP, const double* dN /*[nGP*npe*dm]*/,
                      const double* w /*[nGP]*/, int32_t voigt_kind);
/* material.C copied to every Gauss point by ddsdde_init (stiffnessMtrx.py:124-129) + sigma(F) kind */
int femcy_set_material(femcy_ctx* ctx, int32_t kind, const double* C /*[s*s]*/, const double* params,
                       int32_t nparams);
/* body.get_nodeEles/get_coElement_nodes + sparseIJ (body.py:165-194, stiffnessMtrx.py:70-89):
 * node adjacency -> blocked sliced-ELL matrix, element->slot map, node->element lists */
int femcy_build_pattern(femcy_ctx* ctx);
int femcy_get_pattern_info(femcy_ctx* ctx, femcy_pattern_info* out);
/* which row order femcy_build_pattern took (FEMCY_OPT_NODE_ORDER): used = 0 the caller's numbering, 1 + k = coordinate
 * order k; lines[0] = mean 128-byte cache lines per wavefront gather with the caller's numbering, lines[1 + k] = with
 * coordinate order k (0 where not evaluated) */
int femcy_get_node_order(femcy_ctx* ctx, int32_t* used /* nullable */, double* lines /*[7], nullable*/);

/* ------------------------------------------------------------------------- vector plumbing */
int femcy_vec_upload(femcy_ctx* ctx, int vec, const double* src, int64_t n);
int femcy_vec_download(femcy_ctx* ctx, int vec, double* dst, int64_t n);
int femcy_vec_fill(femcy_ctx* ctx, int vec, double value);                 /* field.fill()          */
int femcy_vec_copy(femcy_ctx* ctx, int dst, int src);                      /* field.copy_from()     */
int femcy_vec_scatter(femcy_ctx* ctx, int vec, const int32_t* idx, const double* vals, int32_t k);
                                                      /* dirichletBC_val, stiffnessMtrx.py:357-366 */
int femcy_vec_sub(femcy_ctx* ctx, int c, int a, int b);                    /* tiGadgets.py:5-9      */
int femcy_vec_axpy(femcy_ctx* ctx, int a, int b, double c, int d);         /* a = b + c*d, :12-16   */
int femcy_vec_scale(femcy_ctx* ctx, int vec, double s);                    /* tiGadgets.py:67-70    */
int femcy_vec_norm(femcy_ctx* ctx, int vec, double* rms);                  /* tiGadgets.py:28-37    */
int femcy_vec_absmax(femcy_ctx* ctx, int vec, double* out);                /* tiGadgets.py:19-25    */

/* --------------------------------------------------------------------------- the hot path */
/* get_dsdx_and_vol + assemble_stiffnessMtrx (stiffnessMtrx.py:132-150, 161-186) at x = X + vec[u];
 * u_vec = -1: the undeformed configuration (u = 0) */
int femcy_assemble_K(femcy_ctx* ctx, int u_vec);
/* assemble_nodal_force_GN (stiffnessMtrx.py:609-644): F (ref. config), sigma(F), dsdx/vol (current
 * config), node-parallel gather of dsdx . sigma * vol */
int femcy_internal_force(femcy_ctx* ctx, int u_vec, int f_vec);
/* One Newton residual evaluation: assemble_nodal_force_GN followed by get_dsdx_and_vol + assemble_stiffnessMtrx on
 * the same displacement (the pair the reference issues back to back at stiffnessMtrx.py:756-759, 779-783 and in
 * inside_relaxation) as ONE element pass: vec[f] = internal force, K = matrix at x = X + vec[u].  Same results as
 * femcy_internal_force + femcy_assemble_K; the second geometry pass and the F / sigma stores are gone (F and sigma
 * "of the last force evaluation" are recomputed when post-processing asks for them). */
int femcy_residual_and_K(femcy_ctx* ctx, int u_vec, int f_vec);
/* dirichletBC_linearEquations (stiffnessMtrx.py:279-307) for one *Boundary block, race-free */
int femcy_apply_dirichlet_linear(femcy_ctx* ctx, const int32_t* dofs, const double* vals, int32_t k, int rhs_vec);
/* dirichletBC_forNewtonMethod_kernel (stiffnessMtrx.py:317-341) */
int femcy_apply_dirichlet_newton(femcy_ctx* ctx, const int32_t* dofs, int32_t k, int residual_vec);
/* Device-resident copy of one *Boundary block's DOF list (node_set*dm + dof).  The reference keeps each node
 * set in a ti.field for the whole run (stiffnessMtrx.py:656-659); the femcy_dofset_* calls are the
 * femcy_apply_dirichlet_* / femcy_vec_scatter calls without the per-call upload and stream sync. */
int femcy_dofset_create(femcy_ctx* ctx, const int32_t* dofs, int32_t k, int32_t* id_out);
int femcy_dofset_dirichlet_newton(femcy_ctx* ctx, int32_t id, int residual_vec);
int femcy_dofset_dirichlet_linear(femcy_ctx* ctx, int32_t id, double value, int rhs_vec);
int femcy_dofset_fill(femcy_ctx* ctx, int32_t id, int vec, double value);       /* dirichletBC_val */
int femcy_dofset_scatter(femcy_ctx* ctx, int32_t id, int vec, const double* vals /*[k]*/);
/* neumannBC (stiffnessMtrx.py:369-411) on the device.  A load set is the device-resident description of one
 * *Dsload surface: for each loaded facet its owning element (body.boundary[facet], body.py:197-216) and its
 * facet type = index into the element plugin's facet tables (facet_natural_coos / facet_point_weights /
 * facet_natural_normals keys, e.g. element_linear_tetrahedral.py:30-55), which are passed as plain arrays:
 *   ft_nodes [nft][nfn]            sorted local node ids of the facet (the dict key)
 *   ft_N     [nft][nip][npe]       shapeFunc at the facet integration points
 *   ft_dN    [nft][nip][npe][dm]   dshape_dnat there
 *   ft_normal[nft][nip][dm]        natural outward normals
 *   ft_weight[nft][nip]            facet point weights
 * femcy_loadset_neumann zero-fills vec[rhs] (reference :384) and writes the consistent nodal loads of
 * traction * (direction, or the outward unit normal n_nat (dx/dxi)^-1 / (|.| + 1e-30) when direction is NULL)
 * on the undeformed geometry; the facet size is |x1 - x0| (dm = 2) or the triangle area of the facet's first
 * three sorted nodes (dm = 3), as ELE.globalNormal computes it.  Sums per node run in a fixed order. */
int femcy_loadset_create(femcy_ctx* ctx, int32_t nft, int32_t nfn, int32_t nip, const int32_t* ft_nodes,
                         const double* ft_N, const double* ft_dN, const double* ft_normal, const double* ft_weight,
                         int32_t nload, const int32_t* load_elem, const int32_t* load_ft, int32_t* id_out);
int femcy_loadset_neumann(femcy_ctx* ctx, int32_t id, double traction, const double* direction /*[dm] or NULL*/,
                          int rhs_vec);
/* compute_Ad (conjugateGradientSolver.py:53-58): vec[y] = K vec[x] */
int femcy_spmv(femcy_ctx* ctx, int x_vec, int y_vec);
/* ConjugateGradientSolver_rowMajor.re_init + solve (conjugateGradientSolver.py:32-51, 103-127):
 * Jacobi-PCG, x0 = 0, stop when max|r| < eps*max|r0|, at most maxit iterations (reference: n). */
int femcy_pcg(femcy_ctx* ctx, int b_vec, int x_vec, double eps, int32_t maxit, int32_t* iters,
              double* rmax0, double* rmax);
/* solve_by_scipy (stiffnessMtrx.py:219-251: `spsolve` on the scipy matrix built from sparseIJ / sparseMtrx_rowMajor,
 * the branch solve_dof takes below 1e5 DOF): vec[x] = K^-1 vec[b] by a direct factorisation on the device.
 * The nodes are renumbered by reverse Cuthill-McKee (once per pattern), K is copied into lower band storage (tiles of
 * 32 x 32, one column panel after the other) and factored K = L S L^T, S = diag(+-1), without pivoting: Cholesky for
 * the positive definite K of a sound configuration (after the Dirichlet treatment, :279-341), and still a
 * factorisation when a diverging Newton iterate has made K indefinite (the reference's LU returns a solution there
 * too, and the increment driver's path depends on it).  Two triangular solves follow; then the residual b - K x is
 * formed with K itself and the solution refined (at most twice) while that pays.  vec[b] is left untouched.
 * FEMCY_ENUMERIC when a pivot is zero / not a number (info->singular_at) or the residual stays above 1e-8 max|b|
 * (elimination without pivoting lost the indefinite matrix): the caller treats it like a solver breakdown.
 * FEMCY_ENOMEM when the band does not fit the limit (FEMCY_OPT_DIRECT_MAX_BYTES).  FEMCY_ECOMM with a communicator
 * attached (the factorisation is single-rank; a partitioned run of a small system keeps the tight PCG of femcy_pcg). */
typedef struct femcy_direct_info {
    int64_t n;               /* scalar unknowns */
    int64_t band_bytes;      /* storage of the band */
    int32_t bandwidth;       /* sub-diagonals kept, in DOF */
    int32_t panels;          /* column panels of 32 DOF (0 on the host backend) */
    int32_t singular_at;     /* 0, or 1 + the row (band order) whose pivot was zero / not a number */
    int32_t negative_pivots; /* 0 = K was positive definite */
    int32_t refinements;     /* refinement steps taken (0 .. 2) */
    int32_t reserved;
    double residual;         /* max|b - K x| / max|b| of the returned x */
} femcy_direct_info;
int femcy_direct_solve(femcy_ctx* ctx, int b_vec, int x_vec, femcy_direct_info* info /* nullable */);
/* What femcy_direct_solve WOULD factor for the current pattern, without factoring: n, bandwidth (after reverse
 * Cuthill-McKee), panels and band_bytes are filled, everything else is zero.  The reference's switch between its two
 * solvers is a fixed DOF count (`solve_dof`, stiffnessMtrx.py:272-276); the host driver here also looks at the band --
 * n * bandwidth^2 flops against iterations * (matrix bytes / bandwidth of the memory) -- before it takes the direct
 * branch (femcy_amd/stiffnessMtrx.py, direct = "auto").  Never FEMCY_ENOMEM: a band beyond the limit is reported, not
 * refused. */
int femcy_direct_plan(femcy_ctx* ctx, femcy_direct_info* info);

/* ------------------------------------------------------------------------ post-processing */
/* compute_strain_stress (stiffnessMtrx.py:436-501): F at vec[u]; strain (infinitesimal, or Green when
 * large != 0); Cauchy stress by constitutiveOfSmallDeform when large == 0 (kept from the last
 * femcy_internal_force otherwise, as in the reference); von Mises stress by material type */
int femcy_compute_strain_stress(femcy_ctx* ctx, int u_vec, int large);
/* get_elasEng (stiffnessMtrx.py:592-606): F at vec[u], elasticEnergyDensity, sum(density * vol) with the
 * vol left by the last geometry pass (reference behaviour) */
int femcy_elastic_energy(femcy_ctx* ctx, int u_vec, double* total);
/* ELE.extrapolate (element_zoo): out[e][a] = sum_g E[a][g] * field[e][g][comp]; E is npe x nGP */
int femcy_extrapolate(femcy_ctx* ctx, int gp_field, int comp, const double* E, double* out /*[ne*npe]*/);

/* --------------------------------------------------------------------- inspection (tests) */
/* the reference's sparseIJ / sparseMtrx_rowMajor layouts (stiffnessMtrx.py:78-94) */
int femcy_get_K_ell(femcy_ctx* ctx, int32_t* ij /*[n*(W+1)]*/, double* A /*[n*W]*/);
/* block-CSR with ascending columns: rowptr[nn+1], colidx[nnzb], vals[nnzb*dm*dm] */
int femcy_get_K_bsr(femcy_ctx* ctx, int32_t* rowptr, int32_t* colidx, double* vals);
int femcy_get_gp_field(femcy_ctx* ctx, int which, double* out);
int femcy_timing(femcy_ctx* ctx, femcy_timing_t* out);
int femcy_timing_reset(femcy_ctx* ctx);

/* ----------------------------------------------------- ceilings of the device (bench.py's roofline) */
/* What the one-launch PCG runs against, measured with the kernel's own launch shape and exchange code:
 * femcy_probe_stream: a read-only sweep of `bytes` (>= 1 MiB), `reps` passes inside one launch after a warm-up
 *   launch, 16-byte loads, XCD-contiguous ranges.  mode 0 = one workgroup of four waves per CU (the persistent
 *   kernel's shape), 1 = the same with non-temporal loads, 2 = eight workgroups per CU, 3 = mode 2 non-temporal.
 *   A buffer below ~200 MB is served by the Infinity Cache from the second pass on, a larger one by HBM.
 *   us_per_pass = microseconds per pass, bytes_per_pass = the bytes one pass reads (whole tiles only).
 * femcy_probe_exchange: one grid-wide exchange of an 8-byte value per workgroup (publish, synchronise, every
 *   workgroup sums all of them), averaged over `rounds` inside one launch.  form 0 = per-XCD counters + top counter
 *   + data (FEMCY_TUNE_PERSIST_VARIANT without bit 1), 1 = tagged 16-byte granules (bit 1).  The sums are checked.
 * femcy_persist_streamed_bytes: bytes of the matrix the persistent PCG streams per iteration on this context (stored
 *   block rows less the register- and LDS-resident ones), 0 when the system does not fit that kernel. */
int femcy_probe_stream(femcy_ctx* ctx, int64_t bytes, int32_t reps, int32_t mode, double* us_per_pass,
                       int64_t* bytes_per_pass /* nullable */);
int femcy_probe_exchange(femcy_ctx* ctx, int32_t rounds, int32_t form, double* us_per_exchange);
/* femcy_probe_spmv: `reps` launches of compute_Ad back to back between ONE pair of HIP events on the context's stream, on
 *   the PCG's own vectors, in the caller's node order (storage_order = 0) or in storage order (1, what the three-launch
 *   PCG of a single rank runs): microseconds from launch to launch = kernel + the boundary between dependent launches. */
int femcy_probe_spmv(femcy_ctx* ctx, int32_t reps, int32_t storage_order, double* us_per_launch);
/* femcy_probe_mailbox (collective, after femcy_comm_mailbox_import): one cross-rank reduction of the persistent
 *   multi-rank PCG -- a wave per rank writes its value into every rank's mailbox and polls its own, the solver's own
 *   code -- averaged over `rounds` inside one launch per rank: the mailbox round trip between the ranks' kernels, link
 *   (xGMI) latency included.  The sums are checked; FEMCY_ECOMM on a time-out. */
int femcy_probe_mailbox(femcy_ctx* ctx, int32_t rounds, double* us_per_round);
int femcy_persist_streamed_bytes(femcy_ctx* ctx, int64_t* bytes);

/* ------------------------------------------------------------------- multi-GPU (new work) */
/* 128-byte ncclUniqueId produced on rank 0 and broadcast by the host program */
int femcy_comm_unique_id(void* id128);
/* 128-byte id of an in-process group instead: the contexts that pass it to femcy_comm_init are driven by one host
 * thread each inside ONE process and exchange through host staging buffers (no RCCL; any devices, also all on
 * one).  Same kernels and call sequence as the RCCL transport -- it is how the multi-rank path is verified on a
 * single GPU.  A rendezvous that is not completed by all ranks within 60 s fails with FEMCY_ECOMM. */
int femcy_comm_local_id(void* id128);
/* 128-byte id of a shared-memory group: the ranks are PROCESSES of one host (any devices, also all on one -- which
 * RCCL refuses) that exchange through a POSIX shared-memory segment; same kernels and call sequence as the other
 * transports.  max_values = the longest vector one collective carries (doubles; the packed interface vector + 8 is
 * enough; at least 4096 are reserved).  Produced by one process and handed to the others by the host program (a
 * pipe, a file, the command line).  It exists so that the cross-process branches of the mailbox path below
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle, kernels of different processes writing into each other's HBM) can run
 * on a one-GPU box; a rendezvous that is not completed within 60 s fails with FEMCY_ECOMM.  Up to 16 ranks. */
int femcy_comm_shm_id(void* id128, int64_t max_values);
/* element partition: this ctx holds one sub-mesh; iface_local_dofs[k] is the local scalar DOF that is
 * entry iface_global_slot[k] of the packed global interface vector (length niface_global), owner[i] = 1
 * if this rank counts local DOF i in dot products (exactly one rank per shared DOF). */
int femcy_comm_init(femcy_ctx* ctx, int32_t rank, int32_t nranks, const void* id128,
                    int32_t niface_local, const int32_t* iface_local_dofs, const int32_t* iface_global_slot,
                    int32_t niface_global, const uint8_t* owner /*[n]*/);
/* With a communicator attached every vector the API hands back is the assembled / replicated one: femcy_spmv,
 * femcy_internal_force and femcy_loadset_neumann sum their sub-assembled result over the interface, femcy_vec_norm
 * (RMS over the DOFs of the whole system) and femcy_vec_absmax count each shared DOF once and reduce over the
 * ranks, femcy_elastic_energy sums over the ranks, femcy_pcg's default iteration cap is the global DOF count.
 * These calls are therefore collective: every rank makes the same sequence of them.
 * femcy_comm_info: rank, ranks and the DOF count of the whole system (owned DOFs summed over the ranks; n and 1
 * rank without a communicator).
 * femcy_iface_sum: sum any other sub-assembled vector over the ranks sharing each interface DOF */
int femcy_comm_info(femcy_ctx* ctx, int32_t* rank, int32_t* nranks, int64_t* n_global);
/* Neighbour exchange, the alternative to the packed all-reduce: nb_rank[k] (ascending) shares the local DOFs
 * nb_dofs[nb_ptr[k] .. nb_ptr[k+1]) with this rank, listed in ascending GLOBAL DOF order on both sides.  Per exchange
 * each pair swaps its segment (ncclSend / ncclRecv in one group) and every interface DOF is summed in ascending rank
 * order, so all replicas get the same bits; d.Ad then travels in an 8-byte all-reduce of its own.  The z-slab
 * partitions have <= 2 neighbours per rank: 2 x 116 KB per exchange at 8 M elements instead of a 0.8 MB all-reduce. */
int femcy_comm_set_neighbours(femcy_ctx* ctx, int32_t nnb, const int32_t* nb_rank, const int32_t* nb_ptr,
                              const int32_t* nb_dofs);
/* collective: times `iters` interface exchanges with each method (the neighbour form including its extra scalar
 * all-reduce), checks that both give the same sums, takes the maximum over the ranks and selects the faster one
 * (FEMCY_OPT_EXCHANGE) on every rank alike.  us[0] / us[1] = microseconds per exchange (all-reduce / neighbour);
 * a failed cross-check keeps the all-reduce and reports us[1] = -1. */
int femcy_comm_tune(femcy_ctx* ctx, int32_t iters, int32_t* chosen, double* us /*[2]*/);
int femcy_iface_sum(femcy_ctx* ctx, int vec);
/* Persistent PCG across ranks (FEMCY_OPT_PCG_PERSIST_MULTI).  After femcy_comm_set_neighbours:
 *   femcy_comm_mailbox_export  allocates this rank's mailbox (fine-grained device memory the peers' kernels write
 *                              into) and describes it in a 256-byte blob: IPC handle, process id, device pointer,
 *                              neighbour segment table;
 *   (the host program hands every rank all blobs, in rank order -- torch.distributed all-gather, MPI, a list)
 *   femcy_comm_mailbox_import  maps the peers' mailboxes (same process: the pointer, with peer access between
 *                              devices; other processes: hipIpcOpenMemHandle) and builds the interface tables;
 *   femcy_comm_persist_agree   collective: enabled = 1 only if every rank can take the path (mailboxes mapped, every
 *                              interface node shared with exactly one other rank as in a slab partition, at most 8
 *                              neighbours, the pattern fits the persistent kernel).
 * femcy_pcg then takes the one-launch path on every rank alike; after each such solve the ranks agree through the
 * communicator whether it completed, and a solve that timed out anywhere is redone everywhere by the RCCL loop. */
int femcy_comm_mailbox_export(femcy_ctx* ctx, void* blob256);
/* collective helper for the step in the middle: `bytes` host bytes of every rank -> recv[nranks][bytes] on every rank,
 * through the context's own transport (no communicator: a copy) -- the host program needs no second communication
 * library to hand the blobs round */
int femcy_comm_allgather_host(femcy_ctx* ctx, const void* send, int32_t bytes, void* recv /*[nranks*bytes]*/);
int femcy_comm_mailbox_import(femcy_ctx* ctx, int32_t nblobs, const void* blobs /*[nranks][256]*/);
int femcy_comm_persist_agree(femcy_ctx* ctx, int32_t* enabled);

#ifdef __cplusplus
}
#endif
#endif /* FEMCY_H */
