// Grid-barrier price on MI355X for the protocols a persistent PCG would use (tools/micro, not part of the library).
//   mode 0: counter barrier only (sc1 arrive / relaxed poll, no fences)           -> us per barrier
//   mode 1: + every WG publishes 8 B (sc1) before and reads all G values after    -> the d.Ad exchange
//   mode 2: + every WG streams `bytes` of a big buffer between barriers           -> barrier under load
//   mode 3: hierarchical: per-XCD counter (blockIdx % 8) then a top counter
//   mode 4: mode 3 + the 8 B exchange (what k_pcg_persist did per reduction at first)
//   mode 5: no counters: every WG publishes its value into a slot of a ring of 3 slot arrays, every WG polls all G
//           slots until none holds the sentinel (a NaN payload no arithmetic produces), and re-arms its slot of the
//           array used two rounds ago -- barrier and exchange in one store latency + one load latency
// usage: barrier_probe G iters bytes_per_wg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ bool barrier_flat(unsigned* counter, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int fail;
    if (threadIdx.x == 0) {
        fail = 0;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { fail = 1; break; }
        }
    }
    __syncthreads();
    return fail == 0;
}

__device__ __forceinline__ bool barrier_xcd(unsigned* xc, unsigned* top, unsigned it, unsigned G) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int fail2;
    if (threadIdx.x == 0) {
        fail2 = 0;
        const unsigned k = blockIdx.x % 8, members = (G + 7 - k) / 8;
        const unsigned prev = __hip_atomic_fetch_add(xc + 32 * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == members * (it + 1))      // last of this XCD group for this round
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned groups = G < 8 ? G : 8;
        unsigned spins = 0;
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups * (it + 1)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { fail2 = 1; break; }
        }
    }
    __syncthreads();
    return fail2 == 0;
}

#define SENTINEL 0x7FF8DEADBEEF0001ull
__device__ __forceinline__ bool is_sentinel(double v) { return (unsigned long long)__double_as_longlong(v) == SENTINEL; }

// all-gather of one double per WG; G <= 256 * SL.  Returns the sum in a fixed order.
template <int SL>
__device__ __forceinline__ double gather_ring(double* ring, unsigned G, int it, double mine, int* fail) {
    __shared__ double sm[4];
    double* cur = ring + (size_t)(it % 3) * G;
    if (threadIdx.x == 0) {
        // re-arm the slot of round it-2 (== it+1 mod 3): every WG has left that round (it published round it-1)
        __hip_atomic_store(ring + (size_t)((it + 1) % 3) * G + blockIdx.x, __longlong_as_double(SENTINEL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cur + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < SL; ++u) {
        const unsigned k = threadIdx.x + 256 * u;
        if (k < G) {
            double v = __hip_atomic_load(cur + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (is_sentinel(v)) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(cur + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1u << 22)) { *fail = 1; break; }
            }
            s += v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// mode 6: the data is the flag (cdna_hip_programming.md Guideline 16, R2): a double travels as two 8-byte granules
// {epoch, half}; the reader re-reads until both tags carry this round's epoch.  No counter, no re-arming.
template <int SL>
__device__ __forceinline__ double gather_tagged(unsigned long long* gran, unsigned G, unsigned epoch, double mine, int* fail) {
    __shared__ double sm[4];
    if (threadIdx.x < 2) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
        const unsigned half = threadIdx.x ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(gran + 2 * blockIdx.x + threadIdx.x, ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < SL; ++u) {
        const unsigned k = threadIdx.x + 256 * u;
        if (k < G) {
            unsigned long long g0, g1;
            unsigned spins = 0;
            for (;;) {
                g0 = __hip_atomic_load(gran + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g1 = __hip_atomic_load(gran + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { *fail = 1; break; }
            }
            s += __longlong_as_double((long long)((g1 << 32) | (g0 & 0xffffffffull)));
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ void __launch_bounds__(256) k_tagged(int iters, unsigned long long* gran, double* out) {
    const unsigned G = gridDim.x;
    __shared__ int fail;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    double acc = 0.0;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        // two arrays alternate: a WG may publish round it+1 while a slower one still reads round it
        unsigned long long* g = gran + (size_t)(it & 1) * 2 * G;
        const double s = G <= 256 ? gather_tagged<1>(g, G, (unsigned)it + 1, (double)it + blockIdx.x, &fail)
                                  : gather_tagged<4>(g, G, (unsigned)it + 1, (double)it + blockIdx.x, &fail);
        const double want = (double)G * it + 0.5 * G * (G - 1.0);
        if (s != want) ++bad;
        acc += s;
        if (fail) break;
    }
    if (threadIdx.x == 0 && (bad || fail)) out[1] = fail ? -1.0 : -2.0;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_ring(int iters, double* ring, double* out) {
    const unsigned G = gridDim.x;
    __shared__ int fail;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    double acc = 0.0;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        const double s = G <= 256 ? gather_ring<1>(ring, G, it, (double)it + blockIdx.x, &fail)
                                  : gather_ring<4>(ring, G, it, (double)it + blockIdx.x, &fail);
        const double want = (double)G * it + 0.5 * G * (G - 1.0);
        if (s != want) ++bad;
        acc += s;
        if (fail) break;
    }
    if (threadIdx.x == 0 && (bad || fail)) out[1] = fail ? -1.0 : -2.0;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_probe(int mode, int iters, unsigned* counters, double* slots, const double2* big,
                                               long per_wg16, double* out) {
    const unsigned G = gridDim.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (mode == 2) {
            const double2* src = big + (long)blockIdx.x * per_wg16;
            for (long i = threadIdx.x; i < per_wg16; i += 256 * 4) {
                double2 a = src[i], b = i + 256 < per_wg16 ? src[i + 256] : a, c = i + 512 < per_wg16 ? src[i + 512] : a,
                        d = i + 768 < per_wg16 ? src[i + 768] : a;
                acc += a.x + b.y + c.x + d.y;
            }
        }
        if (mode >= 1 && mode != 3 && threadIdx.x == 0)
            __hip_atomic_store(slots + (it & 1) * G + blockIdx.x, (double)it + acc * 1e-300, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = mode >= 3 ? barrier_xcd(counters + 64, counters, (unsigned)it, G) : barrier_flat(counters, G * (unsigned)(it + 1));
        if (!ok) { if (threadIdx.x == 0) out[1] = -1.0; return; }
        if (mode >= 1 && mode != 3) {
            double s = 0.0;
            for (unsigned k = threadIdx.x; k < G; k += 256)
                s += __hip_atomic_load(slots + (it & 1) * G + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += s;
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const long bytes = argc > 3 ? atol(argv[3]) : 0;
    unsigned* counters; double *slots, *out; double2* big = nullptr;
    CHECK(hipMalloc(&counters, 4096)); CHECK(hipMalloc(&slots, sizeof(double) * 2 * 4096)); CHECK(hipMalloc(&out, 16));
    const long per16 = bytes / 16;
    if (bytes) { CHECK(hipMalloc(&big, (size_t)G * bytes)); CHECK(hipMemset(big, 0, (size_t)G * bytes)); }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double* ring; CHECK(hipMalloc(&ring, sizeof(double) * 3 * 4096));
    {
        std::vector<unsigned long long> h(3 * 4096, SENTINEL);
        CHECK(hipMemcpy(ring, h.data(), sizeof(double) * 3 * 4096, hipMemcpyHostToDevice));
    }
    for (int mode : {0, 3, 4, 1, 5, 6, 2}) {
        if (mode == 2 && !bytes) continue;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(counters, 0, 4096)); CHECK(hipMemset(out, 0, 16));
            CHECK(hipEventRecord(e0));
            if (mode == 5) {
                std::vector<unsigned long long> h(3 * 4096, SENTINEL);
                CHECK(hipMemcpy(ring, h.data(), sizeof(double) * 3 * 4096, hipMemcpyHostToDevice));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_ring, dim3(G), dim3(256), 0, 0, iters, ring, out);
            } else if (mode == 6) {
                CHECK(hipMemset(ring, 0, sizeof(double) * 3 * 4096));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_tagged, dim3(G), dim3(256), 0, 0, iters, (unsigned long long*)ring, out);
            } else
            hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, mode, iters, counters, slots, big, per16, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double h[2]; CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("G %4d mode %d bytes/WG %8ld: %8.3f us per iteration%s", G, mode, mode == 2 ? bytes : 0, best * 1e3 / iters, h[1] == -1.0 ? "  TIMEOUT\n" : (h[1] == -2.0 ? "  WRONG SUM\n" : "\n"));
        if (mode == 2) printf("      streaming alone would be %.2f us at 6.7 TB/s\n", (double)G * bytes / 6.7e6);
    }
    return 0;
}
