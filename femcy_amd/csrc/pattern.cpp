// Host preprocessing: node adjacency -> blocked SELL-64 layout, element->slot map, per-block
// contribution lists, node->element lists.  Replaces the reference's pure-Python loops
// (body.py:165-194 get_nodeEles/get_coElement_nodes, stiffnessMtrx.py:70-107 sparseIJ/rows/cols),
// which are O(N) interpreter work and would dominate a 1M-element run.  O(ne*npe^2) with small
// constants, threaded over node/element ranges.
#include <algorithm>
#include <cstring>
#include <thread>
#include "ctx.hpp"

namespace femcy {

template <class F>
static void parallel_for(int64_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(hw ? hw : 4, std::max<int64_t>(1, n / 4096));
    nt = std::min(nt, 32);
    if (nt <= 1) {
        f(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    int64_t chunk = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([=] { f(lo, hi, t); });
    }
    for (auto& x : th) x.join();
}

template <class T>
static int upload(T** dptr, const std::vector<T>& h) {
    if (*dptr) {
        (void)hipFree(*dptr);
        *dptr = nullptr;
    }
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    FEMCY_HIP(dmalloc(dptr, bytes));
    if (!h.empty()) FEMCY_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return FEMCY_OK;
}

// SpMV work split: XCD k gets the contiguous slice range holding the k-th eighth of the stored blocks; long
// rows are shared by WPS wavefronts of one workgroup (FEMCY_OPT_SPMV_VARIANT: 0 = choose from the mean row
// length, else 1 / 2 / 4).
void spmv_split(Ctx* c) {
    constexpr int NX = 8, WAVES = 4;
    const int32_t nslices = c->nslices;
    const int64_t stored_rows = c->stored_rows;
    int wps = c->opt_spmv_variant;
    if (wps != 1 && wps != 2 && wps != 4) {
        const double mean_len = nslices ? (double)stored_rows / nslices : 0.0;
        // rocprofv3, C3D10 plate (mean 27 blocks per row): 99 / 88 / 84 us for 1 / 2 / 4 wavefronts per slice;
        // C3D4 (15 blocks per row): equal (round 6, launch to launch: 30.7 / 31.4 / 31.1 us at 1 M elements, 284 / 297 / 307
        // at 8 M -- one wave per slice stays).  2 x 2 blocks (round 6, CPE8 beam of 1 M DOF, 15.6 blocks per row, 36 bytes
        // per block: a wave's share of a slice is half the bytes of a 3 x 3 one): 51.6 / 51.0 / 45.8 us = 0.72 -> 0.81 of
        // HBM with four waves per slice (profiles/r06_spmv_wps.txt)
        wps = c->dm == 2 ? (mean_len > 8.0 ? 4 : 1) : (mean_len <= 20.0 ? 1 : 4);
    }
    c->spmv_wps = wps;
    // matrix stream policy.  Measured (MI355X, C3D4 plate): a 198 MB matrix (1 M elements) is re-read every CG
    // iteration from the 256 MiB Infinity Cache and non-temporal loads cost 16 % (33.1 -> 38.4 us); a 1.55 GB
    // matrix (8 M elements) streams from HBM and non-temporal loads gain 4 % (288 -> 277 us).
    const int64_t matrix_bytes = stored_rows * SLICE * ((int64_t)c->dm * c->dm * 8 + 4);
    const bool beyond_mall = matrix_bytes > (int64_t)256 * 1024 * 1024;
    c->spmv_nt = c->opt_spmv_nt < 0 ? beyond_mall : c->opt_spmv_nt != 0;
    // ... but not all of it: the leading share of every XCD's slice range (235 MB in total) keeps the default policy
    // and is then found in the Infinity Cache by the next product, while the rest streams past it without allocating.
    // Measured per PCG iteration: 124 k C3D10 (379 MB) 100.7 us all non-temporal, 88.3 / 85.9 us with 209 / 247 MB kept,
    // 93.9 with 303 MB; 2 M C3D4 (400 MB) 90.9 -> 86.8 / 87.2 / 87.9 with 180 / 220 / 260 MB kept.
    if (c->opt_spmv_keep >= 0)
        c->spmv_keep_permille = c->opt_spmv_keep;
    else
        c->spmv_keep_permille = matrix_bytes > 0 ? (int)std::min<int64_t>(1000, (int64_t)235 * 1000 * 1000 * 1000 / matrix_bytes) : 0;
    // the PCG vector kernels likewise, but by the size of the VECTORS: 44.7 -> 47.3 us per iteration at 1 M C3D4
    // elements, 84.1 -> 85.1 on the 124 k C3D10 plate (380 MB of matrix, 4.4 MB per vector) and 85.3 -> 87.1 at 2 M
    // C3D4 (8.7 MB per vector): small vectors are re-read from the caches; 176.5 -> 167.0 us at 4 M (17.4 MB per
    // vector), 332.8 -> 312.2 at 8 M
    const bool big_vectors = (int64_t)c->n * 8 > (int64_t)12 * 1000 * 1000;
    c->vec_nt = c->opt_vec_nt < 0 ? (beyond_mall && big_vectors) : c->opt_vec_nt != 0;
    const int spb = WAVES / wps;   // slices per workgroup
    int32_t s = 0;
    c->xcd.start[0] = 0;
    for (int k = 1; k < NX; ++k) {
        const int64_t target = stored_rows * k / NX;
        while (s < nslices && c->h_slice_off[s] < target) ++s;
        c->xcd.start[k] = s;
    }
    c->xcd.start[NX] = nslices;
    int32_t per = 1;
    for (int k = 0; k < NX; ++k) per = std::max(per, (c->xcd.start[k + 1] - c->xcd.start[k] + spb - 1) / spb);
    // Workgroups per XCD.  The product's kernels hold 5 waves per SIMD (84-92 registers), i.e. 160 workgroups per XCD at a
    // time; beyond that the dispatcher hands the next workgroup to whichever CU becomes free, which evens out what the
    // static task lists leave uneven.  Measured in one process, interleaved (round 6, profiles/r06_spmv_rounds.txt): 512
    // instead of 256 -- C3D10 k = 12 (3.0 GB) 597 -> 560 us, k = 8 (0.9 GB) 173 -> 152, 8 M C3D4 (1.55 GB) 283.5 -> 280.5;
    // but twice the d.Ad partials for the vector kernels to sum: the 1 M-DOF CPE8 beam (278 MB) gains 0.5 us in the product
    // and loses 1.3 in the iteration, and at C3D10 k = 6 (357 tasks per XCD) one task per workgroup is the slower form
    // (64.6 -> 65.6): long ranges of large matrices only
    const bool big = per > 512 && matrix_bytes > (int64_t)512 * 1024 * 1024;
    c->spmv_cap_auto = big ? 512 : 256;
    const int32_t bpx = spmv_bpx(per, c->spmv_bpx_cap, c->spmv_cap_auto);
    c->spmv_grid = bpx * NX;                                      // <= 4096 workgroups: larger ranges are looped inside the kernel
    // Rows are sorted by length inside windows of sigma rows, and with task = b + i * bpx (bpx a multiple of the window) a
    // workgroup takes the same position of the window in every round: the head of the window (C3D10: 65 blocks per row
    // against a mean of 29; C3D4: 15 against 14) always on the same workgroups.  Where the lengths spread, the lists are
    // balanced below (mode 64); modes 1 .. 63 rotate the rounds against each other instead (k_spmv).  Interleaved in one
    // process (profiles/r06_spmv_rounds.txt), C3D10 k = 12: 596 us plain at 256 workgroups, 560 plain at 512, 538 rotated,
    // 529 balanced (three-launch iteration 662 -> 597); k = 8: 165 / 144.5 / 147 / 139.  On uniform rows (8 M C3D4) the
    // rotation costs 1.5-2.5 % and the balanced lists are the plain ones: decided per pattern, large matrices only (the
    // 1 M-DOF CPE8 beam and C3D10 at k = 6 keep the plain lists: +-1 us either way)
    if (c->opt_spmv_rot >= 0)
        c->spmv_rot = c->opt_spmv_rot;
    else {
        const int32_t wsl = std::max(1, c->sell_sigma / SLICE);   // slices per sort window
        double spread = 0.0;
        int32_t nw = 0;
        for (int32_t w0 = 0; w0 < nslices; w0 += wsl) {
            const int32_t w1 = std::min(nslices, w0 + wsl);
            int64_t sum = 0;
            int32_t mx = 0;
            for (int32_t q = w0; q < w1; ++q) {
                sum += c->h_slice_len[q];
                mx = std::max(mx, c->h_slice_len[q]);
            }
            if (sum > 0) {
                spread += (double)mx * (w1 - w0) / (double)sum;
                ++nw;
            }
        }
        c->spmv_rot = (big && nw > 0 && spread / nw > 1.25) ? 64 : 0;
    }
    // 64 = lists balanced here instead of rotated blindly: round by round (a round = bpx consecutive tasks, so that the
    // gathers of concurrent workgroups keep sharing the XCD's L2) the workgroup that has collected the most work so far
    // takes the shortest task of the round.  Every workgroup of an XCD then multiplies the same number of stored blocks to
    // within one task.  Deterministic (ties by index); the table is read with one scalar load per round
    c->spmv_perm_rounds = 0;
    if (c->spmv_rot == 64) {
        int32_t rounds = 1;
        for (int k = 0; k < NX; ++k)
            rounds = std::max(rounds, ((c->xcd.start[k + 1] - c->xcd.start[k] + spb - 1) / spb + bpx - 1) / bpx);
        std::vector<int32_t> perm((size_t)NX * rounds * bpx);
        std::vector<int64_t> acc(bpx), wgt(bpx);
        std::vector<int32_t> by_acc(bpx), by_wgt(bpx);
        for (int k = 0; k < NX; ++k) {
            const int32_t s0 = c->xcd.start[k], s1 = c->xcd.start[k + 1];
            std::fill(acc.begin(), acc.end(), 0);
            for (int32_t r = 0; r < rounds; ++r) {
                for (int32_t q = 0; q < bpx; ++q) {
                    int64_t w = 0;
                    for (int32_t u = 0; u < spb; ++u) {
                        const int64_t sl = (int64_t)s0 + ((int64_t)r * bpx + q) * spb + u;
                        if (sl < s1) w += c->h_slice_len[sl];
                    }
                    wgt[q] = w;
                    by_acc[q] = by_wgt[q] = q;
                }
                std::stable_sort(by_wgt.begin(), by_wgt.end(), [&](int32_t a, int32_t b) { return wgt[a] < wgt[b]; });
                std::stable_sort(by_acc.begin(), by_acc.end(), [&](int32_t a, int32_t b) { return acc[a] > acc[b]; });
                int32_t* row = perm.data() + ((size_t)k * rounds + r) * bpx;
                for (int32_t q = 0; q < bpx; ++q) {
                    row[by_acc[q]] = by_wgt[q];
                    acc[by_acc[q]] += wgt[by_wgt[q]];
                }
            }
        }
        if (upload(&c->d_spmv_perm, perm) == FEMCY_OK)
            c->spmv_perm_rounds = rounds;
        else
            c->spmv_rot = 19;                                     // no table: the rotation
    }
}

// Footprints of the storage-order product (k_spmv_fp).  A wave multiplies block rows j0 .. j1 of one slice (its part of
// the slice when long rows are shared by WPS waves); the storage positions those (j1 - j0) x 64 blocks refer to -- its
// footprint -- are listed once, sorted (neighbouring positions: the staging loads coalesce), and the block columns are
// re-expressed as 16-bit indices into that list.  On the 1 M C3D4 plate a wave's 914 references have 437 distinct
// targets in 12 runs; the wave then issues 14 coalesced staging loads instead of 30 gather instructions of 13 lines each.
int ensure_footprint(Ctx* c) {
    const int wps = c->spmv_wps;
    const int64_t key = c->pattern_serial * 8 + wps;
    if (c->fp_serial == key) return FEMCY_OK;
    c->fp_serial = key;
    c->fp_cap = 0;
    const int32_t nslices = c->nslices;
    const int64_t ntask = (int64_t)nslices * wps;
    std::vector<std::vector<int32_t>> lists((size_t)ntask);
    std::vector<uint16_t> lcol((size_t)c->stored_rows * SLICE, 0);
    bool ok = true;
    parallel_for(nslices, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> tmp;
        for (int64_t s = lo; s < hi; ++s) {
            const int32_t L = c->h_slice_len[s];
            const int32_t chunk = (L + wps - 1) / wps;
            const int64_t off = c->h_slice_off[s];
            for (int w = 0; w < wps; ++w) {
                const int32_t j0 = w * chunk, j1 = std::min(L, j0 + chunk);
                tmp.clear();
                for (int32_t j = j0; j < j1; ++j)
                    for (int lane = 0; lane < SLICE; ++lane) tmp.push_back(c->h_pos[c->h_bcol[(off + j) * SLICE + lane]]);
                std::sort(tmp.begin(), tmp.end());
                tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                if (tmp.size() > 65535) ok = false;
                for (int32_t j = j0; j < j1; ++j)
                    for (int lane = 0; lane < SLICE; ++lane) {
                        const int32_t p = c->h_pos[c->h_bcol[(off + j) * SLICE + lane]];
                        lcol[(off + j) * SLICE + lane] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), p) - tmp.begin());
                    }
                lists[(size_t)s * wps + w] = tmp;
            }
        }
    });
    if (!ok) return FEMCY_OK;                                    // fp_cap stays 0: the gather product is used
    std::vector<int32_t> ptr((size_t)ntask + 1, 0);
    size_t cap = 0;
    for (int64_t t = 0; t < ntask; ++t) {
        ptr[t + 1] = ptr[t] + (int32_t)lists[t].size();
        cap = std::max(cap, lists[t].size());
    }
    std::vector<int32_t> fp((size_t)ptr[ntask]);
    for (int64_t t = 0; t < ntask; ++t) std::copy(lists[t].begin(), lists[t].end(), fp.begin() + ptr[t]);
    int rc;
    if ((rc = upload(&c->d_lcol, lcol)) || (rc = upload(&c->d_fp_ptr, ptr)) || (rc = upload(&c->d_fp, fp))) return rc;
    c->fp_cap = (int32_t)((cap + 1) & ~(size_t)1);
    return FEMCY_OK;
}

// Groups of `G` consecutive storage positions (a slice, a chunk of a slice) in Morton order of their centroids: groups
// close in space are then processed close in time and -- with XCD-contiguous ranges of the order -- by the same L2.  What
// it is for: an element's record is needed by every node of the element, i.e. by rows that lie in different groups;
// in storage order (or in the longest-first order of the row-centric assemblies) those fetches are megabytes of other
// records apart and all but the first come over the fabric (CPE8 beam: FETCH 3.7 x the records; C3D10 k = 12, records
// beyond the Infinity Cache: 6-12 x, profiles/r06_pmc_c3d10_k12_first.txt).  Groups of padding rows only sort last.
// Leaves `order` untouched (identity) when the coordinates are not known.
static void spatial_order(const Ctx* c, const std::vector<int32_t>& node_of, int G, std::vector<int32_t>& order) {
    const int dm = c->dm;
    const int64_t ngroups = (int64_t)order.size();
    if ((int64_t)c->h_nodes.size() != (int64_t)c->nn * dm) return;
    double lo[3] = {0, 0, 0}, ext = 0.0;
    for (int d = 0; d < dm; ++d) {
        double mn = c->h_nodes[d], mx = mn;
        for (int32_t a = 1; a < c->nn; ++a) {
            mn = std::min(mn, c->h_nodes[(size_t)a * dm + d]);
            mx = std::max(mx, c->h_nodes[(size_t)a * dm + d]);
        }
        lo[d] = mn;
        ext = std::max(ext, mx - mn);
    }
    const int bits = dm == 3 ? 20 : 30;
    const double scale = ext > 0.0 ? (double)(((int64_t)1 << bits) - 1) / ext : 0.0;
    std::vector<uint64_t> mkey((size_t)ngroups, ~(uint64_t)0);
    parallel_for(ngroups, [&](int64_t glo, int64_t ghi, int) {
        for (int64_t g = glo; g < ghi; ++g) {
            double ctr[3] = {0, 0, 0};
            int cnt = 0;
            for (int r = 0; r < G; ++r) {
                const int32_t a = node_of[(size_t)g * G + r];
                if (a < 0) continue;
                for (int d = 0; d < dm; ++d) ctr[d] += c->h_nodes[(size_t)a * dm + d];
                ++cnt;
            }
            if (!cnt) continue;
            uint64_t k = 0, q[3];
            for (int d = 0; d < dm; ++d) q[d] = (uint64_t)((ctr[d] / cnt - lo[d]) * scale + 0.5);
            for (int b = bits - 1; b >= 0; --b)
                for (int d = dm - 1; d >= 0; --d) k = (k << 1) | ((q[d] >> b) & 1);
            mkey[g] = k;
        }
    });
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return mkey[x] < mkey[y]; });
}

// Pair lists of the FEMCY_ASM_PAIRS assembly: a wavefront owns a chunk of 16 consecutive storage positions (a quarter of
// a slice) and walks the (row, incident element) pairs of its rows, NPE lanes per pair.  The list is in storage order, so
// the kernel's chain is pr_ptr (scalar) -> codes (one coalesced load) -> records, instead of node_of -> ne_ptr -> ne_idx
// -> records.  Order inside a chunk: by row, then ascending element = the summation order of every stored block.
int ensure_pairs(Ctx* c, int RPW, bool spatial, int cpw) {
    const int64_t key = ((c->pattern_serial * 64 + RPW) * 2 + (spatial ? 1 : 0)) * 64 + cpw;
    if (c->pairs_serial == key) return FEMCY_OK;
    const int64_t npos = (int64_t)c->nslices * SLICE;
    const int64_t nchunks = npos / RPW;
    const int dm = c->dm;
    std::vector<int32_t> ptr((size_t)nchunks + 1, 0);
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        int32_t cnt = 0;
        for (int r = 0; r < RPW; ++r) {
            const int32_t a = c->h_node_of[(size_t)ch * RPW + r];
            if (a >= 0) cnt += c->h_ne_ptr[a + 1] - c->h_ne_ptr[a];
        }
        ptr[ch + 1] = ptr[ch] + cnt;
    }
    std::vector<int32_t> code((size_t)ptr[nchunks] + 64, 0);    // + 64 zeros: the list of an empty chunk is read, not used
    parallel_for(nchunks, [&](int64_t lo, int64_t hi, int) {
        for (int64_t ch = lo; ch < hi; ++ch) {
            int32_t w = ptr[ch];
            for (int r = 0; r < RPW; ++r) {
                const int32_t a = c->h_node_of[(size_t)ch * RPW + r];
                if (a < 0) continue;
                for (int32_t k = c->h_ne_ptr[a]; k < c->h_ne_ptr[a + 1]; ++k) {
                    code[w++] = c->h_ne_idx[k] | (r << 27);     // kernels_assembly.hip: PAIR_ROW_SHIFT (the launcher checks ne * npe < 2^27)
                }
            }
        }
    });
    // processing order of the chunks.  A record is needed by every node of its element: npe fetches, from rows that lie
    // in different slices (different mesh lines, different length classes of the sorting windows).  In storage order the
    // fetches of one record are megabytes of other records apart (CPE8 1280 x 128: ~11 MB; an XCD's L2 holds 4) and all but
    // the first come from the Infinity Cache over the fabric: FETCH 3.7 x the records (profiles/r06_pmc_asm_cpe8_pairs.txt).
    // So the chunks are taken in Morton order of their centroids -- chunks close in space run close in time, and with
    // XCD-contiguous ranges of that order on the same L2.
    std::vector<int32_t> order((size_t)nchunks);
    for (int64_t ch = 0; ch < nchunks; ++ch) order[ch] = (int32_t)ch;
    if (spatial) spatial_order(c, c->h_node_of, RPW, order);
    // batches (<= 64 pairs of one chunk) in processing order (kernels_assembly.hip: PairBatch, 32 bytes) and the first
    // batch of every unit of `cpw` chunks (one wavefront's work)
    std::vector<int32_t> desc, unit_ptr;
    desc.reserve((size_t)nchunks * 8);
    for (int64_t k = 0; k < nchunks; ++k) {
        if (k % cpw == 0) unit_ptr.push_back((int32_t)(desc.size() / 8));
        const int32_t ch = order[k];
        const int64_t s = (int64_t)ch * RPW / SLICE;
        const int64_t off = c->h_slice_off[s];
        const int32_t np = ptr[ch + 1] - ptr[ch];
        const int32_t nbat = std::max(1, (np + 63) / 64);
        for (int32_t bb = 0; bb < nbat; ++bb) {
            int32_t d8[8] = {ch, ptr[ch] + 64 * bb, std::min(64, np - 64 * bb), c->h_slice_len[s], 0, 0, bb + 1 == nbat ? 1 : 0, 0};
            memcpy(&d8[4], &off, 8);
            desc.insert(desc.end(), d8, d8 + 8);
        }
    }
    unit_ptr.push_back((int32_t)(desc.size() / 8));
    int rc;
    if ((rc = upload(&c->d_pr_unit, unit_ptr)) || (rc = upload(&c->d_pr_ptr, desc)) || (rc = upload(&c->d_pr_code, code))) return rc;
    c->pairs_serial = key;
    return FEMCY_OK;
}

int build_pattern(Ctx* c) {
    const int32_t nn = c->nn, ne = c->ne, npe = c->npe, dm = c->dm;
    const int32_t* el = c->h_elems.data();
    // the products gather x through buffer resources with 32-bit byte offsets (col * dm * 8; out-of-range buffer loads
    // return 0, i.e. a silently wrong product): a vector must stay below 2 GiB -- 89 M nodes in 3-D, beyond which the
    // matrix alone (> 100 GB) leaves no room for an element pass on one GPU anyway; partition the mesh instead
    FEMCY_REQUIRE(((int64_t)nn + SLICE) * dm * 8 < ((int64_t)1 << 31),
                  "%d nodes x %d DOF exceed the 2 GiB per vector the SpMV gathers address; partition the mesh", nn, dm);

    // ---- node -> incident (element, local index), ascending element order
    std::vector<int32_t> ne_ptr(nn + 1, 0);
    for (int64_t k = 0; k < (int64_t)ne * npe; ++k) ne_ptr[el[k] + 1]++;
    for (int32_t a = 0; a < nn; ++a) ne_ptr[a + 1] += ne_ptr[a];
    std::vector<int32_t> ne_idx((size_t)ne * npe);
    {
        std::vector<int32_t> cur(ne_ptr.begin(), ne_ptr.end() - 1);
        for (int32_t e = 0; e < ne; ++e)
            for (int32_t la = 0; la < npe; ++la) ne_idx[cur[el[(int64_t)e * npe + la]]++] = e * npe + la;
    }
    int32_t max_node_elems = 0;
    for (int32_t a = 0; a < nn; ++a) max_node_elems = std::max(max_node_elems, ne_ptr[a + 1] - ne_ptr[a]);

    // ---- adjacency rows: diagonal first, then ascending neighbours (count pass + fill pass)
    std::vector<int32_t> rowlen(nn, 0);
    parallel_for(nn, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> tmp;
        for (int64_t a = lo; a < hi; ++a) {
            tmp.clear();
            for (int32_t k = ne_ptr[a]; k < ne_ptr[a + 1]; ++k) {
                const int32_t* en = el + (int64_t)(ne_idx[k] / npe) * npe;
                tmp.insert(tmp.end(), en, en + npe);
            }
            tmp.push_back((int32_t)a);   // isolated nodes still own a diagonal block
            std::sort(tmp.begin(), tmp.end());
            rowlen[a] = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        }
    });
    std::vector<int64_t> adj_ptr(nn + 1, 0);
    int32_t max_row = 0;
    for (int32_t a = 0; a < nn; ++a) {
        adj_ptr[a + 1] = adj_ptr[a] + rowlen[a];
        max_row = std::max(max_row, rowlen[a]);
    }
    if (max_row > 65535) {
        set_error("node with %d neighbours exceeds the uint16 slot map", max_row);
        return FEMCY_EINVAL;
    }
    std::vector<int32_t> adj((size_t)adj_ptr[nn]);
    parallel_for(nn, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> tmp;
        for (int64_t a = lo; a < hi; ++a) {
            tmp.clear();
            for (int32_t k = ne_ptr[a]; k < ne_ptr[a + 1]; ++k) {
                const int32_t* en = el + (int64_t)(ne_idx[k] / npe) * npe;
                tmp.insert(tmp.end(), en, en + npe);
            }
            tmp.push_back((int32_t)a);
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            int32_t* row = adj.data() + adj_ptr[a];
            int32_t w = 0;
            row[w++] = (int32_t)a;
            for (int32_t b : tmp)
                if (b != a) row[w++] = b;
        }
    });

    // ---- SELL-C-sigma: inside windows of `sigma` nodes, rows are stored in order of decreasing length, so the 64
    // rows of a slice have (nearly) equal length and the padding disappears (C3D10: 16 % -> < 2 %); the window keeps
    // a slice's nodes spatially close, so the x-gathers stay local.  pos[a] = storage position of node a.
    //
    // Round 4: the windows run over an internal node ORDER, not necessarily the caller's numbering.  With the solver
    // vectors in storage order (kernels_pcg.hip) a wave's gather of block column j reads positions pos[col(lane, j)]:
    // what it costs is the number of cache lines those 64 addresses touch.  Lanes are rows of one length class in
    // ascending order; if the order walks the mesh along lines (a lexicographic coordinate order on a structured
    // mesh) their j-th neighbours are consecutive members of THEIR class, i.e. consecutive positions: 14 lines per
    // gather on the C3D10 plate instead of 27 (caller's numbering: corners first, then mid-side nodes by edge);
    // Morton and reverse Cuthill-McKee orders give 31 (tools/gather_lines.py) -- locality is not the point,
    // regularity is.  So the candidates (the caller's numbering and the lexicographic orders of the quantised
    // coordinates, one per axis permutation) are MEASURED on the pattern -- mean lines per gather over sampled slices --
    // and the best one is taken if it beats the caller's numbering by 10 %.
    const int32_t nslices = (nn + SLICE - 1) / SLICE;
    const int32_t sigma = std::max(SLICE, (c->sell_sigma / SLICE) * SLICE);
    std::vector<int32_t> node_of((size_t)nslices * SLICE, -1), pos(nn);
    auto layout = [&](const std::vector<int32_t>* order, std::vector<int32_t>& node_of_, std::vector<int32_t>& pos_) {
        parallel_for((nn + sigma - 1) / sigma, [&](int64_t lo, int64_t hi, int) {
            std::vector<int32_t> idx;
            for (int64_t w = lo; w < hi; ++w) {
                const int32_t a0 = (int32_t)(w * sigma), a1 = std::min(nn, a0 + sigma);
                idx.resize(a1 - a0);
                for (int32_t a = a0; a < a1; ++a) idx[a - a0] = order ? (*order)[a] : a;
                std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return rowlen[x] > rowlen[y]; });
                for (int32_t k = 0; k < a1 - a0; ++k) {
                    node_of_[a0 + k] = idx[k];
                    pos_[idx[k]] = a0 + k;
                }
            }
        });
    };
    // mean number of 128-byte lines one wave gather touches (dm doubles per lane at pos * dm * 8), over <= 192 slices
    auto gather_lines = [&](const std::vector<int32_t>& node_of_, const std::vector<int32_t>& pos_) -> double {
        const int32_t full = nn / SLICE;                         // slices without padding lanes
        if (full < 1) return 0.0;
        const int32_t nsamp = std::min<int32_t>(full, 192);
        int64_t lines = 0, gathers = 0;
        std::vector<int64_t> ln(2 * SLICE);
        for (int32_t t = 0; t < nsamp; ++t) {
            const int32_t s = (int32_t)((int64_t)t * full / nsamp);
            int32_t L = 0;
            for (int lane = 0; lane < SLICE; ++lane) L = std::max(L, rowlen[node_of_[(size_t)s * SLICE + lane]]);
            for (int32_t j = 0; j < L; ++j) {
                for (int lane = 0; lane < SLICE; ++lane) {
                    const int32_t a = node_of_[(size_t)s * SLICE + lane];
                    const int32_t col = j < rowlen[a] ? adj[adj_ptr[a] + j] : a;
                    const int64_t b0 = (int64_t)pos_[col] * dm * 8;
                    ln[2 * lane] = b0 >> 7;
                    ln[2 * lane + 1] = (b0 + dm * 8 - 1) >> 7;
                }
                std::sort(ln.begin(), ln.end());
                lines += std::unique(ln.begin(), ln.end()) - ln.begin();
                ++gathers;
            }
        }
        return gathers ? (double)lines / (double)gathers : 0.0;
    };
    c->node_order_used = 0;
    for (double& v : c->node_order_cost) v = 0.0;
    layout(nullptr, node_of, pos);
    // (the measured choice needs a few full slices to measure on; a forced order is always applied)
    if (c->opt_node_order >= 1 && (int64_t)c->h_nodes.size() == (int64_t)nn * dm && (c->opt_node_order >= 2 || nn >= 4 * SLICE)) {
        const double natural = gather_lines(node_of, pos);
        c->node_order_cost[0] = natural;
        // coordinates quantised on one common scale (21 bits per axis): grid lines of a structured mesh compare equal
        double lo[3] = {0, 0, 0}, ext = 0.0;
        for (int d = 0; d < dm; ++d) {
            double mn = c->h_nodes[d], mx = mn;
            for (int32_t a = 1; a < nn; ++a) {
                mn = std::min(mn, c->h_nodes[(size_t)a * dm + d]);
                mx = std::max(mx, c->h_nodes[(size_t)a * dm + d]);
            }
            lo[d] = mn;
            ext = std::max(ext, mx - mn);
        }
        const double scale = ext > 0.0 ? (double)((1 << 21) - 1) / ext : 0.0;
        std::vector<uint32_t> q((size_t)nn * dm);
        for (int32_t a = 0; a < nn; ++a)
            for (int d = 0; d < dm; ++d) q[(size_t)a * dm + d] = (uint32_t)((c->h_nodes[(size_t)a * dm + d] - lo[d]) * scale + 0.5);
        static const int perms3[6][3] = {{2, 1, 0}, {2, 0, 1}, {1, 2, 0}, {1, 0, 2}, {0, 2, 1}, {0, 1, 2}};   // slowest .. fastest
        static const int perms2[2][3] = {{1, 0, 0}, {0, 1, 0}};
        const int ncand = dm == 3 ? 6 : 2;
        double best = natural;
        int best_k = -1;
        std::vector<int32_t> order(nn), best_node_of, best_pos, cn((size_t)nslices * SLICE, -1), cp(nn);
        std::vector<std::pair<uint64_t, int32_t>> keyed(nn);
        for (int k = 0; k < ncand; ++k) {
            if (c->opt_node_order >= 2 && c->opt_node_order - 2 != k) continue;
            const int* pm = dm == 3 ? perms3[k] : perms2[k];
            for (int32_t a = 0; a < nn; ++a) {
                uint64_t key = 0;
                for (int d = 0; d < dm; ++d) key = (key << 21) | q[(size_t)a * dm + pm[d]];
                keyed[a] = {key, a};
            }
            std::sort(keyed.begin(), keyed.end());                 // ties (coincident nodes) by node number
            for (int32_t a = 0; a < nn; ++a) order[a] = keyed[a].second;
            std::fill(cn.begin(), cn.end(), -1);
            layout(&order, cn, cp);
            const double cost = gather_lines(cn, cp);
            c->node_order_cost[1 + k] = cost;
            if (c->opt_node_order >= 2 || cost < best) {
                best = cost;
                best_k = k;
                best_node_of = cn;
                best_pos = cp;
            }
        }
        if (best_k >= 0 && (c->opt_node_order >= 2 || best < 0.9 * natural)) {
            node_of.swap(best_node_of);
            pos.swap(best_pos);
            c->node_order_used = 1 + best_k;
        }
    }
    std::vector<int32_t> slice_len(nslices, 0);
    std::vector<int64_t> slice_off(nslices + 1, 0);
    for (int32_t s = 0; s < nslices; ++s) {
        int32_t L = 0;
        for (int32_t lane = 0; lane < SLICE; ++lane) {
            const int32_t a = node_of[(size_t)s * SLICE + lane];
            if (a >= 0) L = std::max(L, rowlen[a]);
        }
        slice_len[s] = L;
        slice_off[s + 1] = slice_off[s] + L;
    }
    const int64_t stored_rows = slice_off[nslices];
    if (stored_rows * SLICE >= (int64_t)INT32_MAX) {
        set_error("pattern too large for 32-bit block positions (%lld stored blocks)", (long long)stored_rows * SLICE);
        return FEMCY_EINVAL;
    }
    std::vector<int32_t> bcol((size_t)stored_rows * SLICE);
    parallel_for(nslices, [&](int64_t lo, int64_t hi, int) {
        for (int64_t s = lo; s < hi; ++s)
            for (int32_t j = 0; j < slice_len[s]; ++j)
                for (int32_t lane = 0; lane < SLICE; ++lane) {
                    const int32_t a = node_of[s * SLICE + lane];
                    int32_t col = 0;
                    if (a >= 0) col = (j < rowlen[a]) ? adj[adj_ptr[a] + j] : a;   // padding: zero block on own node
                    bcol[(slice_off[s] + j) * SLICE + lane] = col;
                }
    });

    // ---- element (la, lb) -> slot j in the row of node a
    const int64_t npair = (int64_t)ne * npe * npe;
    if (npair >= (int64_t)INT32_MAX) {
        set_error("ne*npe^2 = %lld exceeds 32-bit contribution codes", (long long)npair);
        return FEMCY_EINVAL;
    }
    std::vector<uint16_t> slotj((size_t)npair);
    std::vector<int32_t> ctr_cnt((size_t)stored_rows * SLICE + 1, 0);
    auto block_pos = [&](int32_t a, int32_t j) -> int64_t {
        return (slice_off[pos[a] / SLICE] + j) * SLICE + (pos[a] % SLICE);
    };
    parallel_for(ne, [&](int64_t lo, int64_t hi, int) {
        for (int64_t e = lo; e < hi; ++e)
            for (int32_t la = 0; la < npe; ++la) {
                int32_t a = el[e * npe + la];
                const int32_t* row = adj.data() + adj_ptr[a];
                for (int32_t lb = 0; lb < npe; ++lb) {
                    int32_t b = el[e * npe + lb];
                    int32_t j = 0;
                    if (b != a) j = (int32_t)(std::lower_bound(row + 1, row + rowlen[a], b) - row);
                    slotj[(e * npe + la) * npe + lb] = (uint16_t)j;
                }
            }
    });
    for (int64_t k = 0; k < npair; ++k) {
        int64_t e = k / ((int64_t)npe * npe);
        int32_t la = (int32_t)((k / npe) % npe);
        ctr_cnt[block_pos(el[e * npe + la], slotj[k]) + 1]++;
    }
    for (size_t p = 1; p < ctr_cnt.size(); ++p) ctr_cnt[p] += ctr_cnt[p - 1];
    std::vector<int32_t> ctr((size_t)npair);
    {
        std::vector<int32_t> cur(ctr_cnt.begin(), ctr_cnt.end() - 1);
        for (int64_t k = 0; k < npair; ++k) {   // ascending (e, la, lb): fixed summation order
            int64_t e = k / ((int64_t)npe * npe);
            int32_t la = (int32_t)((k / npe) % npe);
            ctr[cur[block_pos(el[e * npe + la], slotj[k])]++] = (int32_t)k;
        }
    }

    // ---- transposed-block map for the symmetric gather assembly
    std::vector<int32_t> tpos((size_t)stored_rows * SLICE, -1);
    parallel_for(nn, [&](int64_t lo, int64_t hi, int) {
        for (int64_t a = lo; a < hi; ++a) {
            const int32_t* row = adj.data() + adj_ptr[a];
            tpos[block_pos((int32_t)a, 0)] = (int32_t)block_pos((int32_t)a, 0);
            for (int32_t j = 1; j < rowlen[a]; ++j) {
                const int32_t b = row[j];
                if (b < a) {
                    tpos[block_pos((int32_t)a, j)] = -2;
                } else {
                    const int32_t* rb = adj.data() + adj_ptr[b];
                    const int32_t jb = (int32_t)(std::lower_bound(rb + 1, rb + rowlen[b], (int32_t)a) - rb);
                    tpos[block_pos((int32_t)a, j)] = (int32_t)block_pos(b, jb);
                }
            }
        }
    });

    // ---- launch order of the row-centric assembly (one workgroup per slice): slices by decreasing WORK, which for that
    // kernel is (row, incident element) pairs plus a per-row constant -- a C3D10 corner row has 24 elements, a mid-side
    // row 4-8.  Workgroups are dispatched in index order and land on XCD b % 8, so longest-first (a) hands every XCD an
    // even share of every weight class whatever the row order is, (b) lets the short slices fill the tail.
    std::vector<int32_t> asm_order(nslices);
    {
        std::vector<int64_t> work(nslices, 0);
        for (int32_t s = 0; s < nslices; ++s) {
            asm_order[s] = s;
            for (int lane = 0; lane < SLICE; ++lane) {
                const int32_t a = node_of[(size_t)s * SLICE + lane];
                if (a >= 0) work[s] += (ne_ptr[a + 1] - ne_ptr[a]) + 2;
            }
        }
        std::stable_sort(asm_order.begin(), asm_order.end(), [&](int32_t x, int32_t y) { return work[x] > work[y]; });
    }
    // ... and by locality (round 6): the same kernel on a mesh whose element records exceed the Infinity Cache is bound by
    // re-fetching them -- every record once per node of its element, from whichever XCD the row's slice landed on.
    // Slices in Morton order of their centroids, XCD-contiguous ranges of that order (the kernel's `xcdc` mapping).
    std::vector<int32_t> asm_order_near(nslices);
    for (int32_t s = 0; s < nslices; ++s) asm_order_near[s] = s;
    spatial_order(c, node_of, SLICE, asm_order_near);

    // ---- commit to the context
    c->nslices = nslices;
    c->stored_rows = stored_rows;
    c->nnzb = adj_ptr[nn];
    c->max_row_blocks = max_row;
    c->max_node_elems = max_node_elems;
    c->h_slice_len = slice_len;
    c->h_slice_off.assign(slice_off.begin(), slice_off.end());
    c->h_rowlen = rowlen;
    c->h_pos = pos;
    c->h_bcol = bcol;
    c->h_node_of = node_of;
    c->h_ne_ptr = ne_ptr;
    c->h_ne_idx = ne_idx;
    spmv_split(c);

    int rc;
    if ((rc = upload(&c->d_slice_len, slice_len))) return rc;
    if ((rc = upload(&c->d_slice_off, c->h_slice_off))) return rc;
    if ((rc = upload(&c->d_rowlen, rowlen))) return rc;
    if ((rc = upload(&c->d_pos, pos))) return rc;
    if ((rc = upload(&c->d_node_of, node_of))) return rc;
    if ((rc = upload(&c->d_bcol, bcol))) return rc;
    if ((rc = upload(&c->d_slotj, slotj))) return rc;
    if ((rc = upload(&c->d_ctr_ptr, ctr_cnt))) return rc;
    if ((rc = upload(&c->d_ctr, ctr))) return rc;
    if ((rc = upload(&c->d_tpos, tpos))) return rc;
    if ((rc = upload(&c->d_ne_ptr, ne_ptr))) return rc;
    if ((rc = upload(&c->d_ne_idx, ne_idx))) return rc;
    if ((rc = upload(&c->d_asm_order, asm_order))) return rc;
    if ((rc = upload(&c->d_asm_order_near, asm_order_near))) return rc;
    {
        std::vector<int32_t> ident(nslices);
        for (int32_t s = 0; s < nslices; ++s) ident[s] = s;
        if ((rc = upload(&c->d_asm_order_id, ident))) return rc;
    }

    if (c->d_Kvals) (void)hipFree(c->d_Kvals);
    size_t kbytes = (size_t)stored_rows * dm * dm * SLICE * sizeof(double);
    FEMCY_HIP(dmalloc(&c->d_Kvals, std::max<size_t>(kbytes, 8)));
    FEMCY_HIP(dfill_sync(c->d_Kvals, 0, kbytes));       // landed on return (ctx.hpp: the round-5 NaN was this fill running late)
    return FEMCY_OK;
}

}  // namespace femcy
