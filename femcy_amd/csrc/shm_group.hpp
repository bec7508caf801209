// Shared-memory group: the rendezvous of N PROCESSES of one host (round 4).  One POSIX shared-memory segment holds a
// header (generation barrier on lock-free atomics, per-rank neighbour tables) and one staging area per rank.  No HIP
// in this file: comm.cpp copies device data into / out of the staging areas; tests/native/shm_group_test.cpp drives
// the same code from host arrays in the CPU suite (rendezvous, sums in rank order, time-out of a missing rank).
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

namespace femcy {

static const char SHM_MAGIC[8] = {'F', 'E', 'M', 'C', 'Y', 'S', 'H', 'M'};
constexpr int SHM_MAXR = 16;

struct ShmHeader {
    std::atomic<uint32_t> ready;                 // set last by the creating rank
    uint32_t nranks;
    uint64_t cap;                                // doubles per staging area
    std::atomic<uint32_t> joined, left, arrived, gen, broken;
    uint32_t pad_;
    struct Rank {
        int64_t count;
        int32_t nnb;
        int32_t nb_rank[SHM_MAXR];
        int32_t nb_ptr[SHM_MAXR + 1];
    } rk[SHM_MAXR];
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "the shared-memory group needs lock-free 32-bit atomics");

inline uint64_t process_nonce() {                // identifies THIS process (pids repeat across pid namespaces)
    static const uint64_t nonce = [] {
        std::random_device rd;
        uint64_t v = ((uint64_t)rd() << 32) ^ rd();
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        v ^= (uint64_t)ts.tv_nsec * 0x9E3779B97F4A7C15ull ^ ((uint64_t)getpid() << 17);
        return v ? v : 1;
    }();
    return nonce;
}

inline void shm_make_id(void* id128, int64_t cap_doubles) {
    static std::atomic<uint64_t> counter{1};
    std::memset(id128, 0, 128);
    std::memcpy(id128, SHM_MAGIC, 8);
    const uint64_t token = process_nonce() ^ (counter.fetch_add(1) * 0xD6E8FEB86659FD93ull);
    const uint64_t cap = (uint64_t)(cap_doubles > 4096 ? cap_doubles : 4096);
    std::memcpy((char*)id128 + 8, &token, 8);
    std::memcpy((char*)id128 + 16, &cap, 8);
}

struct ShmGroup {
    ShmHeader* h = nullptr;
    double* stage = nullptr;                     // [nranks][cap]
    size_t bytes = 0;
    char name[64] = {0};
    double timeout_s = 60.0;                     // a rendezvous that takes longer fails (a rank died)
    std::string err;

    double* area(int r) const { return stage + (size_t)r * h->cap; }

    static void nap(unsigned spins) {
        if (spins < 2000) {
            sched_yield();
        } else {
            timespec ts{0, 50000};
            nanosleep(&ts, nullptr);
        }
    }
    static double now_s() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    bool fail(const char* what) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "shared-memory group %s: %s", name, what);
        err = buf;
        return false;
    }

    // all ranks arrive, or the call fails after timeout_s (and the group stays broken for everybody)
    bool barrier() {
        if (h->broken.load(std::memory_order_acquire)) return fail("a previous rendezvous failed");
        const uint32_t gen0 = h->gen.load(std::memory_order_acquire);
        if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == h->nranks) {
            h->arrived.store(0, std::memory_order_relaxed);
            h->gen.fetch_add(1, std::memory_order_release);
            return true;
        }
        const double t0 = now_s();
        for (unsigned spins = 0;; ++spins) {
            if (h->gen.load(std::memory_order_acquire) != gen0) return true;
            if (h->broken.load(std::memory_order_acquire) || now_s() - t0 > timeout_s) {
                h->broken.store(1, std::memory_order_release);
                return fail("rendezvous timed out (a rank is missing)");
            }
            nap(spins);
        }
    }

    // join the group the id names (rank 0 creates the segment)
    bool open(int32_t rank, int32_t nranks, const void* id128) {
        if (nranks < 1 || nranks > SHM_MAXR || rank < 0 || rank >= nranks) return fail("1 .. 16 ranks");
        uint64_t token, cap;
        std::memcpy(&token, (const char*)id128 + 8, 8);
        std::memcpy(&cap, (const char*)id128 + 16, 8);
        std::snprintf(name, sizeof(name), "/femcy_%016llx", (unsigned long long)token);
        const size_t head = (sizeof(ShmHeader) + 4095) & ~(size_t)4095;
        bytes = head + sizeof(double) * (size_t)cap * nranks;
        // rank 0 creates the segment, everybody else opens it without O_CREAT: a rank that arrives after the group has
        // already failed and been unlinked (leave() below) must not found a second, split group under the same name and
        // sit in it until the rendezvous times out -- it fails here with "the segment did not appear"
        const bool creator = rank == 0;
        int fd = creator ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : -1;
        if (creator && fd < 0) return fail("the segment exists already (a stale /dev/shm entry, or two ranks 0)");
        if (!creator) {
            const double t0 = now_s();
            for (unsigned spins = 0; (fd = shm_open(name, O_RDWR, 0600)) < 0; ++spins) {
                if (now_s() - t0 > timeout_s) return fail("the segment did not appear");
                nap(spins);
            }
        }
        if (creator && ftruncate(fd, (off_t)bytes) != 0) {
            close(fd);
            shm_unlink(name);
            return fail("ftruncate failed");
        }
        if (!creator) {                                  // the creator may still be sizing the segment
            const double t0 = now_s();
            struct stat st;
            for (unsigned spins = 0; fstat(fd, &st) != 0 || (size_t)st.st_size < bytes; ++spins) {
                if (now_s() - t0 > timeout_s) {
                    close(fd);
                    return fail("the segment never reached its size");
                }
                nap(spins);
            }
        }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) {
            if (creator) shm_unlink(name);
            return fail("mmap failed");
        }
        h = reinterpret_cast<ShmHeader*>(m);
        stage = reinterpret_cast<double*>((char*)m + head);
        if (creator) {                                   // a fresh segment is zero-filled
            h->nranks = (uint32_t)nranks;
            h->cap = cap;
            h->ready.store(1, std::memory_order_release);
        } else {
            const double t0 = now_s();
            for (unsigned spins = 0; !h->ready.load(std::memory_order_acquire); ++spins) {
                if (now_s() - t0 > timeout_s) {
                    munmap(m, bytes);
                    h = nullptr;
                    return fail("the segment was never initialised");
                }
                nap(spins);
            }
        }
        if (h->nranks != (uint32_t)nranks || h->cap != cap) {
            munmap(m, bytes);
            h = nullptr;
            return fail("rank count / capacity do not match the group");
        }
        // once everybody has mapped the segment its name is not needed any more: unlinked then, so that nothing stays
        // behind in /dev/shm when a rank dies later
        if (h->joined.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) shm_unlink(name);
        return true;
    }

    void leave() {
        if (!h) return;
        // a rank that leaves before everybody joined (an error during set-up) removes the name itself -- the last joiner
        // would have -- and marks the group broken first, so that the ranks already inside fail at their next rendezvous
        // at once instead of after the time-out
        if (h->joined.load(std::memory_order_acquire) < h->nranks) {
            h->broken.store(1, std::memory_order_release);
            shm_unlink(name);
        }
        h->left.fetch_add(1, std::memory_order_acq_rel);
        munmap((void*)h, bytes);
        h = nullptr;
    }

    // sum (gather = false) or concatenation (true) of `count` doubles per rank, in rank order: the same bits everywhere.
    // `mine` may be the rank's own staging area (then nothing is copied in)
    bool exchange(int rank, const double* mine, int64_t count, double* out, bool gather) {
        if ((uint64_t)count > h->cap) return fail("more values than the staging area holds (femcy_comm_shm_id)");
        if (mine != area(rank)) std::memcpy(area(rank), mine, sizeof(double) * (size_t)count);
        h->rk[rank].count = count;
        if (!barrier()) return false;
        for (int r = 0; r < (int)h->nranks; ++r)
            if (h->rk[r].count != count) return fail("the ranks passed different lengths to one collective");
        if (gather) {
            for (int r = 0; r < (int)h->nranks; ++r) std::memcpy(out + (size_t)r * count, area(r), sizeof(double) * (size_t)count);
        } else {
            for (int64_t i = 0; i < count; ++i) out[i] = 0.0;
            for (int r = 0; r < (int)h->nranks; ++r) {
                const double* src = area(r);
                for (int64_t i = 0; i < count; ++i) out[i] += src[i];
            }
        }
        return barrier();                                // everyone has read every staging area
    }
};

}  // namespace femcy
