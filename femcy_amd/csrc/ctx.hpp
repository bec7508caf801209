// Internal state of one femcy context (one HIP device + one stream).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/femcy.h"

namespace femcy {

void set_error(const char* fmt, ...);

#define FEMCY_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            femcy::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return FEMCY_EHIP;                                                                  \
        }                                                                                       \
    } while (0)

// Every device allocation of the library goes through dmalloc.  FEMCY_DEBUG_POISON=1 (environment, read once) fills
// each one with 0xFF bytes (a NaN as f64, -1 as i32) before it is used: a kernel that reads memory nobody wrote -- which a
// fresh process hides, because fresh device memory is zero, and a long-lived one does not, because the allocator hands
// freed blocks back -- then fails deterministically: the race / uninitialised-read check of the -m gpu suite
// (tools/records/r05_gpu2.sh).  In that mode a fill that cannot be issued or completed is an allocation failure, not a
// silently unpoisoned buffer.
// A fill of device memory that HAS LANDED when the call returns.  hipMemset on the null stream is asynchronous for device
// memory and the context streams are non-blocking (they do not order behind the null stream): a zero-fill issued by
// femcy_build_pattern could still be running when the first assembly kernel of the context stream wrote the same
// buffer, and zeroed what the kernel had just written -- the "NaN after 1 iteration" of round 5 (a K with 160 empty rows ->
// 1 / 0 in the Jacobi vector), reproduced and bisected in round 6 (profiles/r06_nan_hunt.txt).  The fill runs on a
// non-blocking stream of its own ON THE DEVICE OF THE ALLOCATION (several ranks of one process hold one device each; a
// device-wide synchronisation would deadlock against their co-dependent persistent kernels) and is synchronised here.
inline hipError_t dfill_sync(void* p, int value, size_t bytes) {
    if (!bytes) return hipSuccess;
#ifdef FEMCY_TEST_LATE_FILL     /* the behaviour before the fix, for tests/test_gpu_regressions.py to be shown failing on */
    return hipMemset(p, value, bytes);
#endif
    constexpr int MAXDEV = 64;
    static hipStream_t fill_stream[MAXDEV] = {};
    static std::mutex mu;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && (dev < 0 || dev >= MAXDEV)) e = hipErrorInvalidDevice;
    hipStream_t st = nullptr;
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(mu);
        if (!fill_stream[dev]) e = hipStreamCreateWithFlags(&fill_stream[dev], hipStreamNonBlocking);
        st = fill_stream[dev];
    }
    if (e == hipSuccess) e = hipMemsetAsync(p, value, bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

inline hipError_t dmalloc(void** p, size_t bytes) {
    static const bool poison = [] {
        const char* e = getenv("FEMCY_DEBUG_POISON");
        return e && e[0] && e[0] != '0';
    }();
    const hipError_t rc = hipMalloc(p, bytes);
    if (rc != hipSuccess || !poison || !bytes) return rc;
    // (the fill must have LANDED before the caller's first copy into the new buffer: index arrays of -1 and a memory fault
    // that was the checker's own)
    const hipError_t e = dfill_sync(*p, 0xFF, bytes);
    if (e != hipSuccess) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    return e;
}
template <class T>
inline hipError_t dmalloc(T** p, size_t bytes) {
    return dmalloc(reinterpret_cast<void**>(p), bytes);
}

#define FEMCY_REQUIRE(cond, ...)                    \
    do {                                            \
        if (!(cond)) {                              \
            femcy::set_error(__VA_ARGS__);          \
            return FEMCY_EINVAL;                    \
        }                                           \
    } while (0)

// FEMCY_TUNE_PAIRS default: XCD-contiguous (1) + 8 rows per wave (2) + Morton chunk order (32) + 3 chunks per wave (128):
// 113-115 us on the 1 M-DOF CPE8 beam against 126 with all bits clear (profiles/r06_asm_cpe8_knobs.txt)
constexpr int FEMCY_PAIRS_DEFAULT = 163;
constexpr int SLICE = 64;           // nodes per SELL slice = one wavefront
constexpr int MAX_PARTIALS = 4096;  // upper bound on per-launch reduction partials

// Position of entry k (= r*dm + c) of the dm x dm block stored in block-row `row` for lane `lane`.
// Entries are interleaved in pairs, [row][k/2][lane][k%2] with the odd last entry (dm = 3: k = 8) as a
// trailing [lane] plane, so that a lane reads/writes two consecutive doubles per instruction: a wavefront
// moves 1 KiB per 16-byte load (4 + one 8-byte load per 3x3 block instead of 9 8-byte loads).
template <int DM>
__host__ __device__ __forceinline__ int64_t kv_index(int64_t row, int k, int lane) {
    constexpr int DD = DM * DM, NP = DD / 2;
    const int64_t base = row * (int64_t)(DD * SLICE);
    return (k < 2 * NP) ? base + (k >> 1) * (2 * SLICE) + lane * 2 + (k & 1) : base + NP * (2 * SLICE) + lane;
}
inline int64_t kv_index_rt(int dm, int64_t row, int k, int lane) {
    return dm == 3 ? kv_index<3>(row, k, lane) : kv_index<2>(row, k, lane);
}

// device-side scalar state of a PCG solve (conjugateGradientSolver.py:103-127)
struct PcgState {
    double rMr[2];   // r.M.r, double-buffered by iteration parity
    double r0;       // max|r0|
    double rmax;     // max|r| after the last completed iteration
    double dAd;      // multi-rank: reduced d.Ad
    double alpha;    // step length of the running iteration (k_update_xr -> k_update_d)
    double eps;      // stopping tolerance of this solve (kept on the device so that kernel arguments are
                     // solve-independent and a burst of iterations can be replayed from a hipGraph)
    int32_t iters;   // completed iterations (written by k_update_d, read by k_update_xr of the next iteration)
    int32_t it_k3;   // iteration index handed from k_update_xr to k_update_d
    int32_t done;    // 0 running, 1 converged, 2 NaN/breakdown: written by k_update_d, tested by k_spmv / k_update_xr
    int32_t skip;    // `done` as k_update_xr saw it, handed to the k_update_d of the same iteration
    unsigned long long xround;   // launches of k_update_fused so far (its granule tag); never reset by a solve
};

// contiguous slice range of each XCD for the SpMV (balanced by stored blocks), passed by value
struct XcdRanges {
    int32_t start[9];
};
// workgroups per XCD of a product whose longest XCD range holds `per` tasks: `cap` (the knob) or `cap_auto` (spmv_split)
inline int32_t spmv_bpx(int32_t per, int32_t cap, int32_t cap_auto) {
    if (cap <= 0) cap = cap_auto;
    return per < 1 ? 1 : (per < cap ? per : cap);
}

struct EventPair {
    hipEvent_t a, b;
};

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;

    // ---- mesh / element / material
    int32_t nn = 0, dm = 0, ne = 0, npe = 0, nGP = 0, voigt = 0, s = 0;
    int64_t n = 0;
    std::vector<int32_t> h_elems;
    double* d_nodes = nullptr;
    int32_t* d_elems = nullptr;
    double* d_dN = nullptr;
    double* d_w = nullptr;
    double* d_C = nullptr;
    int32_t mat_kind = -1;
    double mat_params[4] = {0, 0, 0, 0};
    double h_C[36] = {0};             // host copy of C (the consistent tangent reads lambda, mu from it)
    bool C_is_cubic = false;          // dm = 3: C has the cubic pattern (c11, c12, c44; kblock_cubic3)
    double cubic[3] = {0, 0, 0};
    bool have_mesh = false, have_element = false, have_material = false, have_pattern = false;
    bool dN_sums_to_zero = false;     // element tables satisfy sum_a dN_a = 0 (partition of unity)

    // ---- blocked SELL-64 matrix (lane = node, diagonal block in slot 0)
    int32_t nslices = 0;
    int32_t* d_asm_order = nullptr;   // row-centric assembly (rows4): slices by decreasing work (workgroup b takes entry b)
    int32_t* d_asm_order_near = nullptr;   // ... or in Morton order of their centroids, taken in XCD-contiguous ranges (records
                                      // beyond the Infinity Cache: FEMCY_TUNE_ROWS4_ORDER)
    int32_t* d_asm_order_id = nullptr;     // ... or in storage order (experiments)
    int tune_rows4_order = -1;        // -1 auto (by the size of the element records), 0 longest first, 1 locality
    XcdRanges xcd{};                  // SpMV: slice range per XCD
    int32_t spmv_grid = 0;            // 8 * max blocks per XCD
    int32_t spmv_wps = 1;             // wavefronts per slice (1, 2 or 4)
    bool spmv_nt = false;             // matrix stream with non-temporal loads (matrix larger than the Infinity Cache)
    bool vec_nt = false;              // PCG vector kernels with non-temporal accesses
    int opt_vec_nt = -1;              // -1 auto (same rule as the matrix stream), 0 / 1 forced (test knob 103)
    int opt_spmv_nt = -1;             // -1 auto, 0 / 1 forced (test knob 102)
    int32_t spmv_bpx_cap = 0;         // SpMV workgroups per XCD (larger slice ranges are looped in the kernel); 0 = by the
                                      // size of the matrix (spmv_cap_auto), FEMCY_TUNE_SPMV_WG_PER_XCD fixes it
    int32_t spmv_cap_auto = 256;      // spmv_split: 512 for long ranges of a large matrix, else 256
    int32_t opt_spmv_rot = -1;        // FEMCY_TUNE_SPMV_ROT: -1 = by the spread of the row lengths (spmv_split), 64 = balanced lists
    int32_t spmv_rot = 0;             // rotation of the product's rounds against each other (k_spmv), 0 = none
    int32_t* d_spmv_perm = nullptr;   // [8][spmv_perm_rounds][workgroups per XCD] balanced task lists (spmv_split), or nullptr
    int32_t spmv_perm_rounds = 0;
    int64_t stored_rows = 0;          // sum over slices of slice_len (in block rows of 64 lanes)
    int64_t nnzb = 0;
    int32_t max_row_blocks = 0, max_node_elems = 0;
    std::vector<int32_t> h_slice_len, h_rowlen, h_bcol, h_pos;
    std::vector<int64_t> h_slice_off;
    int32_t* d_slice_len = nullptr;
    int64_t* d_slice_off = nullptr;
    int32_t* d_rowlen = nullptr;      // [nn] blocks per node (indexed by node)
    int32_t* d_pos = nullptr;         // [nn] node -> storage position (slice*64 + lane): SELL-C-sigma row order
    int32_t* d_node_of = nullptr;     // [nslices*64] storage position -> node, -1 for padding lanes
    int32_t sell_sigma = 4096;        // sorting window (nodes); 64 = natural order.  (32768 measured -1 % ... +1 % per PCG
                                      // iteration inside bench.py on the C3D4 and C3D10 plates: no reason to move it)
    int32_t* d_bcol = nullptr;        // [stored_rows*64]
    double* d_Kvals = nullptr;        // [stored_rows*dm*dm*64]
    uint16_t* d_slotj = nullptr;      // [ne*npe*npe] element-local (a,b) -> slot j in row of node a
    int32_t* d_ctr_ptr = nullptr;     // [stored_rows*64+1] contributions per stored block
    int32_t* d_ctr = nullptr;         // [ne*npe*npe] packed (e*npe+la)*npe+lb
    int32_t* d_tpos = nullptr;        // [stored_rows*64] position of the transposed block (b,a) if a < b; p on the
                                      // diagonal; -2 if a > b (the mirror lane stores it); -1 padding
    int32_t* d_ne_ptr = nullptr;      // [nn+1] node -> incident elements
    int32_t* d_ne_idx = nullptr;      // [ne*npe] packed e*npe+la
    // pair lists of FEMCY_ASM_PAIRS (pattern.cpp: ensure_pairs, built on first use): per chunk of 16 (or 8) consecutive storage
    // positions the (row, incident element) pairs in (row, ascending element) order -- code e*npe+la and row inside the chunk
    std::vector<int32_t> h_node_of, h_ne_ptr, h_ne_idx;
    int32_t* d_pr_ptr = nullptr;      // PairBatch descriptors (32 B) in PROCESSING order
    int32_t* d_pr_unit = nullptr;     // [units + 1] first batch of every wavefront's unit of chunks
    int32_t* d_pr_code = nullptr;     // [ne*npe + 64] row inside the chunk << 27 | e*npe+la
    int64_t pairs_serial = -1;        // pattern_serial * 64 + rows per chunk the lists were built for
    int tune_pairs = FEMCY_PAIRS_DEFAULT;   // FEMCY_TUNE_PAIRS (kernels_assembly.hip)

    // ---- Gauss-point fields
    double* d_dsdx = nullptr;
    double* d_vol = nullptr;
    double* d_F = nullptr;
    double* d_sigma = nullptr;
    double* d_strain = nullptr;
    double* d_mises = nullptr;
    double* d_energy = nullptr;
    double* d_fe = nullptr;           // [ne][npe][dm] per-element nodal forces of the last femcy_internal_force
    // The force evaluation of a Newton residual does not store F and sigma (at 1 M C3D4 they are 143 of the 334 MB the
    // element pass would write, and nothing on the solve path reads them).  The reference's post-processing reads
    // "the stress of the last force evaluation", so that state is kept as a copy of the displacement it was made
    // with (8 n bytes) and F / sigma are recomputed from it when somebody asks (ensure_gp_stress).
    bool gp_lazy = false;
    double* d_u_lazy = nullptr;

    // ---- vectors
    double* d_vec[FEMCY_VEC_COUNT] = {nullptr};
    double *d_r = nullptr, *d_d = nullptr, *d_M = nullptr, *d_Ad = nullptr;
    double* d_part1 = nullptr;        // [MAX_PARTIALS] d.Ad partials (SpMV launch)
    double* d_part2 = nullptr;        // [2*MAX_PARTIALS] (r.M.r, max|r|) partials
    PcgState* d_state = nullptr;
    PcgState* h_state = nullptr;      // pinned
    char* h_stage = nullptr;          // pinned staging area for small host -> device payloads (stage_h2d)
    size_t stage_cap = 0, stage_used = 0;
    double* h_scalar = nullptr;       // pinned scratch for reductions
    int32_t* d_idx_scratch = nullptr;
    double* d_val_scratch = nullptr;
    int64_t scratch_cap = 0;

    // ---- hipGraph of one poll-burst of PCG iterations (single rank, timing off)
    hipGraphExec_t pcg_graph = nullptr;
    const double* pcg_graph_x = nullptr;
    int pcg_graph_iters = 0, pcg_graph_g = 0, pcg_graph_np1 = 0;
    int opt_graph = 1;
    // ---- persistent one-launch PCG for systems of one wavefront-task per SIMD (k_pcg_persist)
    int opt_persist = 1;              // FEMCY_OPT_PCG_PERSIST
    // limit on the STREAMED part of the matrix in the persistent PCG.  Rounds 2-4: 240 MiB (the Infinity Cache), from a
    // round-2 measurement (124 k C3D10, 280 MB streamed: 99-103 us here against 93 with three launches).  Re-measured in
    // round 5 with the round-4 kernel (nt stream, tagged granules, storage-order d, 16 + 8 byte gathers): 61.0 us
    // against 78.0 (profiles/r05_persist_hbm_c3d10.txt) -- the rule was stale.  What bounds the kernel is the vector
    // layout (<= 4 slices per wave), not the matrix: no byte limit by default any more (test knob: option 114)
    int64_t persist_max_bytes = (int64_t)1 << 40;
    int tune_rows4_tile = 0;          // FEMCY_TUNE_ROWS4_TILE: 1000 GP + LCUT, 0 = off (round-5 experiment, kernels_assembly.hip)
    int opt_persist_rj = 4;           // block rows per slice kept in registers (test knob 105)
    int opt_persist_wgs = 0;          // test knob 107: workgroups of the launch (0 = one per CU; more than that cannot
                                      // be co-resident, the barrier times out and the solve falls back)
    int opt_persist_l2rows = 1;       // FEMCY_TUNE_PERSIST_L2_ROWS: default-policy streamed rows per slice (VAR & 1)
    int opt_persist_variant = -1;     // FEMCY_TUNE_PERSIST_VARIANT bits (-1 = defaults)
    int opt_persist_dbg = 0;          // timing experiments only (test knob 106): skip parts of the iteration
    int opt_persist_lds = -1;         // block rows per wave kept in LDS (-1: as many as fit; test knob 104)
    int persist_cus = 0;              // compute units of the device
    bool persist_failed = false;      // a grid barrier timed out once: do not try again on this context
    bool small_failed = false;        // the same for k_pcg_small
    int opt_skip_occupancy = 0;       // FEMCY_TUNE_SKIP_OCCUPANCY_CHECK
    uint32_t barrier_spin_limit = 1u << 20;   // polls (s_sleep 1 each, ~0.3 us) before a grid barrier gives up: ~0.5 s
    int32_t* d_persist_assign = nullptr;   // [waves][SPW] slices of each wave (LPT-balanced per XCD range), -1 = none
    std::vector<int64_t> persist_assign_key;
    int64_t pattern_serial = 0;       // bumped by femcy_build_pattern
    double* d_persist = nullptr;      // d double buffer, partials, granules, barrier counters
    int64_t persist_cap = 0;
    int32_t* d_bcolp = nullptr;       // block columns as storage positions (PCG with its vectors in storage order)
    int64_t bcolp_serial = -1;
    // ---- footprint product (k_spmv_fp): per (slice, wave part) the sorted list of storage positions its block rows
    // refer to, and the block columns as 16-bit indices into that list; the wave stages x of its footprint in LDS with
    // coalesced loads and gathers from there
    int opt_spmv_fp = 0;              // FEMCY_OPT_SPMV_FOOTPRINT (0 off, 1 on where the footprints fit the LDS)
    uint16_t* d_lcol = nullptr;       // [stored_rows * 64]
    int32_t* d_fp_ptr = nullptr;      // [nslices * wps + 1]
    int32_t* d_fp = nullptr;
    int32_t fp_cap = 0;               // longest footprint (entries), 0 = not usable
    int64_t fp_serial = -1;           // pattern_serial * 8 + wps the arrays were built for
    int opt_fused_update = 0;         // FEMCY_OPT_PCG_FUSED_UPDATE: single-rank three-launch loop with ONE vector kernel per iteration
                                      // (measured slower than the two kernels: default off)
    bool fused_failed = false;        // its in-kernel exchange timed out once: two kernels from then on
    double* d_fused = nullptr;        // granules of k_update_fused ([1024][2] x 16 B)
    int opt_pos_space = 1;            // FEMCY_OPT_PCG_STORAGE_ORDER: the three-kernel PCG of a single rank keeps r, d, M, Ad, x
                                      // in storage order (gathers of neighbouring lanes then hit neighbouring addresses)
    double* d_posb = nullptr;         // right-hand side / solution in storage order
    double* d_posx = nullptr;
    int64_t pos_cap = 0;
    int opt_node_order = 1;           // FEMCY_OPT_NODE_ORDER: 0 = rows sorted inside windows of the caller's numbering,
                                      // 1 (default) = inside windows of the best of a few coordinate orders if it beats the
                                      // caller's numbering by 10 % (measured on the pattern), 2 + k = forced: coordinate order k
    int node_order_used = 0;          // 0 natural, 1 + k = coordinate order k (femcy_pattern_info)
    double node_order_cost[8] = {0};  // mean 128-byte lines per wave gather, natural first (diagnostics)
    std::vector<double> h_nodes;      // host copy of the coordinates (femcy_set_mesh) for the ordering
    char* d_probe = nullptr;          // femcy_probe_stream's buffer
    int64_t probe_cap = 0;
    // ---- one-launch PCG for small systems (k_pcg_small)
    int spmv_keep_permille = 0;       // NT SpMV: share of every XCD's slice range kept on the default cache policy
    int opt_spmv_keep = -1;           // -1 auto (235 MB of the matrix), else per mille (tuning knob 110)
    int opt_small_rr = -1;            // test knob 108: block rows per wave of the small-system PCG kept in registers
    int opt_small = 1;                // FEMCY_OPT_PCG_SMALL
    int small_max_lds = 65536;        // LDS a workgroup may allocate (device attribute, femcy_ctx_create)
    int small_max_wg = 128;           // workgroups that are certainly co-resident at one per CU
    double* d_small = nullptr;        // Ad double buffer, d.Ad partials, barrier counter
    int64_t small_cap = 0;
    int ew_cap = 512;                 // FEMCY_OPT_EW_GRID

    // ---- device-resident DOF lists of *Boundary blocks
    struct DofSet { int32_t* d_dofs; double* d_vals; int32_t k; };
    std::vector<DofSet> dofsets;

    // ---- device-resident *Dsload surfaces (femcy_loadset_*)
    struct LoadSet {
        int32_t nft, nfn, nip, nload, nnode;
        int32_t *d_ft_nodes, *d_elem, *d_ft, *d_node, *d_ptr, *d_slot;
        double *d_N, *d_dN, *d_normal, *d_weight, *d_contrib, *d_dir;
    };
    std::vector<LoadSet> loadsets;

    // ---- options / timing
    int opt_assembly = FEMCY_ASM_AUTO;
    int opt_tangent = 0;              // FEMCY_OPT_TANGENT
    int opt_poll = 32;
    int opt_timing = 0;               // 0 off, 1 every launch, k > 1: every k-th SpMV launch
    int64_t spmv_count = 0;
    int opt_timing_fence = 1;
    int opt_spmv_variant = 0;
    femcy_timing_t timing{};
    std::vector<EventPair> ev_pool;
    struct Pending { int cls; size_t ev; };
    std::vector<Pending> ev_pending;
    size_t ev_next = 0;

    // ---- multi-rank
    int32_t rank = 0, nranks = 1;
    int64_t n_global = 0;             // DOFs of the un-partitioned system (sum of owned DOFs over the ranks)
    void* comm = nullptr;             // ncclComm_t, the in-process group (comm_local) or the shared-memory group
    int comm_kind = 0;                // COMM_*
    bool comm_local = false;
    uint64_t comm_token = 0;
    int32_t niface_local = 0, niface_global = 0;
    int32_t* d_iface_dof = nullptr;
    int32_t* d_iface_slot = nullptr;
    int32_t* d_slot2dof = nullptr;    // [niface_global] local DOF of each global interface slot, -1 if not held
    uint8_t* d_owner = nullptr;
    double* d_commbuf = nullptr;      // [niface_global + 8]
    double* d_gather = nullptr;       // [nranks*2]

    // ---- persistent PCG across ranks (kernels_pcg_persist.hip "persistent PCG across ranks"): this rank's mailbox,
    // the peers' mailboxes as mapped on this device, the per-position interface table
    unsigned long long* d_mbox = nullptr;
    int64_t mbox_words = 0;
    bool mbox_finegrained = false;
    std::vector<unsigned long long*> h_peer_mbox;   // [nranks], own entry = d_mbox
    std::vector<void*> ipc_opened;                  // mappings to close
    unsigned long long** d_peer_tab = nullptr;      // device copy of h_peer_mbox
    int32_t* d_mr_tab = nullptr;                    // [nslices * 64][4]
    std::vector<int32_t> h_nb_dofs;                 // host copy of the neighbour DOF lists
    bool persist_multi_local = false;               // this rank could take the path
    bool persist_multi = false;                     // ... and every rank agreed (femcy_comm_persist_agree)
    bool persist_multi_failed = false;              // a solve timed out: the RCCL loop for the next solves
    int persist_multi_fallbacks = 0;                // ... counted here; the one-launch path is retried after PERSIST_MULTI_RETRY
    int opt_persist_multi = 1;                      // FEMCY_OPT_PCG_PERSIST_MULTI
    uint32_t solve_serial = 0;

    // ---- overlapped iteration (neighbour exchange): interface slices first, their exchange on a second stream
    // while the interior slices are multiplied
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_iface = nullptr, ev_xchg = nullptr;
    int opt_overlap = 1;              // FEMCY_OPT_OVERLAP
    bool split_ready = false;
    std::vector<int32_t> h_iface_dof; // host copy of the local interface DOFs (femcy_comm_init)
    int32_t* d_split_list = nullptr;  // [nslices] slices holding an interface node first, then the others
    int32_t n_if_slices = 0;

    // ---- neighbour exchange (femcy_comm_set_neighbours): the alternative to the packed all-reduce
    int exchange = 0;                 // FEMCY_OPT_EXCHANGE: 0 packed all-reduce, 1 neighbour send/recv
    std::vector<int32_t> h_nb_rank, h_nb_ptr;   // neighbours (ascending) and their segments of the send / recv buffers
    int32_t* d_nb_dofs = nullptr;     // [nb_total] local DOF of every send entry
    double* d_nb_send = nullptr;      // [nb_total]
    double* d_nb_recv = nullptr;      // [nb_total]
    int32_t* d_if_ptr = nullptr;      // [niface_local+1] per local interface DOF (order of d_iface_dof): its
    int32_t* d_if_src = nullptr;      //   contributions in ascending rank order; src = recv index, -1 = own value
};

enum { COMM_NONE = 0, COMM_RCCL = 1, COMM_LOCAL = 2, COMM_SHM = 3 };
// timing classes
enum { T_GEOM = 0, T_ASM = 1, T_FORCE = 2, T_SPMV = 3, T_PCG = 4, T_PERSIST = 5 };
size_t timing_begin(Ctx* c, int cls);
EventPair* timing_acquire(Ctx* c, int cls);   // registers a pair without recording (hipExtLaunchKernel fills it)
void timing_end(Ctx* c, size_t h);
void timing_collect(Ctx* c);

// pattern.cpp
int build_pattern(Ctx* c);
void spmv_split(Ctx* c);
// kernels_*.hip (host launchers)
// what the element pass leaves in memory: current-configuration gradients + det J w (always computed), the
// deformation gradient, the Cauchy stress, the per-element nodal forces
enum : unsigned { GEOM_DSDX = 1, GEOM_F = 2, GEOM_SIGMA = 4, GEOM_FE = 8 };
int launch_geom(Ctx* c, const double* d_u, unsigned what);
int ensure_gp_stress(Ctx* c);   // F / sigma of the last force evaluation, recomputed on demand (see Ctx::gp_lazy)
int launch_post(Ctx* c, int large);
int launch_energy(Ctx* c);
int launch_extrapolate(Ctx* c, const double* d_E, const double* d_field, int width, int comp, double* d_out);
int launch_energy_sum(Ctx* c, double* total);
int launch_assemble(Ctx* c);
int launch_nodal_force(Ctx* c, double* d_f);
int launch_neumann(Ctx* c, const Ctx::LoadSet& ls, double traction, bool along_normal, double* d_rhs);
// pos_space: x and y are in STORAGE order (entry p * dm + c belongs to the node at storage position p = slice * 64 + lane;
// the padding lanes of the last slice hold zeros) -- the form the three-kernel PCG runs in since round 4
int launch_spmv(Ctx* c, const double* d_x, double* d_y, double* d_partials, int* nblocks_out, bool pos_space = false);
// part 1 / 2 of the split product: the slices holding interface nodes / all others (partials go to
// d_partials[part_off ...]); split_prepare builds the slice list once per pattern + communicator
int launch_spmv_part(Ctx* c, int part, const double* d_x, double* d_y, double* d_partials, int part_off, int* nblocks_out);
int split_prepare(Ctx* c);
bool coresident(Ctx* c, const void* fn, int block, size_t lds, int grid);
int probe_stream(Ctx* c, int64_t bytes, int32_t reps, int32_t mode, double* us_per_pass, int64_t* bytes_per_pass);
int probe_exchange(Ctx* c, int32_t rounds, int32_t form, double* us_per_exchange);
int probe_mailbox(Ctx* c, int32_t rounds, double* us_per_round);
int probe_spmv(Ctx* c, int32_t reps, int32_t storage_order, double* us_per_launch);
int spmv_public_storage_order(Ctx* c, const double* d_x, double* d_y);   // femcy_spmv through the storage-order kernel
int64_t persist_streamed_bytes(Ctx* c);
int ensure_pairs(Ctx* c, int rows_per_chunk, bool spatial_order, int chunks_per_wave);   // pattern.cpp: d_pr_unit / d_pr_ptr / d_pr_code for the current pattern
int ensure_footprint(Ctx* c);   // pattern.cpp: d_lcol / d_fp_ptr / d_fp for the current pattern and spmv_wps
int ensure_pos_vectors(Ctx* c);   // d_posb / d_posx (storage-order right-hand side / solution) + d_bcolp
int ensure_bcolp(Ctx* c);   // d_bcolp = pos[bcol]: block columns as storage positions
int pcg_persist_solve(Ctx* c, const double* d_b, double* d_x, double eps, int32_t maxit, bool* handled);
int launch_dirichlet_zero(Ctx* c, const int32_t* d_dofs, int32_t k, double* d_resid_or_null);
int pcg_solve(Ctx* c, const double* d_b, double* d_x, double eps, int32_t maxit, int32_t* iters, double* r0,
              double* rmax);
int vec_fill(Ctx* c, double* d, double v, int64_t n);
int vec_sub(Ctx* c, double* dc, const double* da, const double* db);
int vec_axpy(Ctx* c, double* da, const double* db, double cc, const double* dd);
int vec_scale(Ctx* c, double* d, double s);
int vec_sumsq(Ctx* c, const double* d, double* out);
int vec_absmax(Ctx* c, const double* d, double* out);
int vec_scatter(Ctx* c, double* d, const int32_t* d_idx, const double* d_vals, int32_t k);
int vec_scatter_const(Ctx* c, double* d, const int32_t* d_idx, double val, int32_t k);
int ensure_scratch(Ctx* c, int64_t k);
// comm.cpp
int comm_unique_id(void* id128);
int comm_local_id(void* id128);
int comm_shm_id(void* id128, int64_t cap_doubles);
int comm_allgather_host(Ctx* c, const void* send, int32_t bytes, void* recv);
int comm_init(Ctx* c, int32_t rank, int32_t nranks, const void* id128);
int comm_allreduce_sum(Ctx* c, double* d_buf, int64_t count);
int comm_allgather(Ctx* c, const double* d_send, double* d_recv, int64_t count);
int comm_neighbour_exchange(Ctx* c, hipStream_t stream);   // d_nb_send segments -> neighbours, their segments -> d_nb_recv
int comm_register_neighbours(Ctx* c);  // in-process transport: publish this rank's segment table
int comm_destroy(Ctx* c);
int comm_mailbox_export(Ctx* c, void* blob256);
int comm_mailbox_import(Ctx* c, int32_t nblobs, const void* blobs);
int comm_persist_agree(Ctx* c, int32_t* enabled);
bool persist_pattern_fits(Ctx* c);
int iface_sum(Ctx* c, double* d_v);
int scalar_across_ranks(Ctx* c, double* d_val, int mode, double* out);
void pcg_graph_reset(Ctx* c);

}  // namespace femcy

struct femcy_ctx {
    femcy::Ctx c;
};
