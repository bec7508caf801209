// C entry points over femcy_amd/csrc/shm_group.hpp for the CPU suite (tests/test_shm_group_cpu.py): the rendezvous of
// the shared-memory transport driven from host arrays, no HIP.  Test infrastructure, not part of the product library.
#include "../../femcy_amd/csrc/shm_group.hpp"

extern "C" {

void shmtest_make_id(void* id128, int64_t cap) { femcy::shm_make_id(id128, cap); }

void* shmtest_open(int rank, int nranks, const void* id128, double timeout_s) {
    auto* g = new femcy::ShmGroup();
    g->timeout_s = timeout_s;
    if (!g->open(rank, nranks, id128)) {
        std::fprintf(stderr, "shmtest_open: %s\n", g->err.c_str());
        delete g;
        return nullptr;
    }
    return g;
}

int shmtest_exchange(void* h, int rank, const double* in, int64_t n, double* out, int gather) {
    return static_cast<femcy::ShmGroup*>(h)->exchange(rank, in, n, out, gather != 0) ? 0 : -1;
}

const char* shmtest_error(void* h) { return static_cast<femcy::ShmGroup*>(h)->err.c_str(); }
const char* shmtest_name(void* h) { return static_cast<femcy::ShmGroup*>(h)->name; }

void shmtest_leave(void* h) {
    auto* g = static_cast<femcy::ShmGroup*>(h);
    g->leave();
    delete g;
}
}
