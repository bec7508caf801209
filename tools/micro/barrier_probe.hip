// Grid-barrier price on MI355X for the protocols a persistent PCG would use (tools/micro, not part of the library).
//   mode 0: counter barrier only (sc1 arrive / relaxed poll, no fences)           -> us per barrier
//   mode 1: + every WG publishes 8 B (sc1) before and reads all G values after    -> the d.Ad exchange
//   mode 2: + every WG streams `bytes` of a big buffer between barriers           -> barrier under load
//   mode 3: hierarchical: per-XCD counter (blockIdx % 8) then a top counter
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
        bool ok = mode == 3 ? barrier_xcd(counters + 64, counters, (unsigned)it, G) : barrier_flat(counters, G * (unsigned)(it + 1));
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
    for (int mode : {0, 3, 1, 2}) {
        if (mode == 2 && !bytes) continue;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(counters, 0, 4096)); CHECK(hipMemset(out, 0, 16));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, mode, iters, counters, slots, big, per16, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double h[2]; CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("G %4d mode %d bytes/WG %8ld: %8.3f us per iteration%s", G, mode, mode == 2 ? bytes : 0, best * 1e3 / iters, h[1] < 0 ? "  TIMEOUT\n" : "\n");
        if (mode == 2) printf("      streaming alone would be %.2f us at 6.7 TB/s\n", (double)G * bytes / 6.7e6);
    }
    return 0;
}
