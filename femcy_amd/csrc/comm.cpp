// RCCL plumbing for the element-partitioned solve.  The reference is single-device (SURVEY.md 2:
// "Collectives: none"); this is new work.  One process per GPU; each rank holds a sub-mesh whose
// sub-assembled K has partial sums on interface rows.  Per CG iteration there are two collectives:
//   1. all-reduce(sum, f64) of the packed global interface vector of y = K_loc d with the local
//      d.K_loc.d appended (d^T A d = sum_ranks d_r^T K_r d_r, so no owner mask is needed there),
//   2. all-gather of the per-rank (r.M.r over owned DOFs, max|r|) pair.
// RCCL is bound with dlopen so that libfemcy_hip.so shares whatever librccl the host process already
// loaded (PyTorch bundles its own copy with the same SONAME) and single-GPU users need none.
#include <dlfcn.h>
#include <cstring>
#include "ctx.hpp"

namespace femcy {

typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(void**, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

static struct {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
} R;

static int load_rccl() {
    if (R.lib) return FEMCY_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        R.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // prefer an already-loaded copy
        if (R.lib) break;
    }
    for (const char* nm : names) {
        if (R.lib) break;
        R.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!R.lib) {
        set_error("cannot load librccl: %s", dlerror());
        return FEMCY_ECOMM;
    }
    R.get_uid = (fn_get_uid)dlsym(R.lib, "ncclGetUniqueId");
    R.init_rank = (fn_init_rank)dlsym(R.lib, "ncclCommInitRank");
    R.allreduce = (fn_allreduce)dlsym(R.lib, "ncclAllReduce");
    R.allgather = (fn_allgather)dlsym(R.lib, "ncclAllGather");
    R.destroy = (fn_destroy)dlsym(R.lib, "ncclCommDestroy");
    R.errstr = (fn_errstr)dlsym(R.lib, "ncclGetErrorString");
    if (!R.get_uid || !R.init_rank || !R.allreduce || !R.allgather || !R.destroy) {
        set_error("librccl is missing required symbols");
        return FEMCY_ECOMM;
    }
    return FEMCY_OK;
}

#define FEMCY_NCCL(call)                                                                       \
    do {                                                                                       \
        int _r = (call);                                                                       \
        if (_r != 0) {                                                                         \
            set_error("%s failed: %s", #call, R.errstr ? R.errstr(_r) : "rccl error");         \
            return FEMCY_ECOMM;                                                                \
        }                                                                                      \
    } while (0)

int comm_unique_id(void* id128) {
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid id;
    FEMCY_NCCL(R.get_uid(&id));
    std::memcpy(id128, &id, sizeof(id));
    return FEMCY_OK;
}

int comm_init(Ctx* c, int32_t rank, int32_t nranks, const void* id128) {
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid id;
    std::memcpy(&id, id128, sizeof(id));
    FEMCY_HIP(hipSetDevice(c->device));
    FEMCY_NCCL(R.init_rank(&c->comm, nranks, id, rank));
    c->rank = rank;
    c->nranks = nranks;
    return FEMCY_OK;
}

int comm_allreduce_sum(Ctx* c, double* d_buf, int64_t count) {
    if (!c->comm || count <= 0) return FEMCY_OK;
    FEMCY_NCCL(R.allreduce(d_buf, d_buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->comm, c->stream));
    return FEMCY_OK;
}

int comm_allgather(Ctx* c, const double* d_send, double* d_recv, int64_t count) {
    if (!c->comm) {
        FEMCY_HIP(hipMemcpyAsync(d_recv, d_send, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return FEMCY_OK;
    }
    FEMCY_NCCL(R.allgather(d_send, d_recv, (size_t)count, /*ncclFloat64*/ 8, c->comm, c->stream));
    return FEMCY_OK;
}

int comm_destroy(Ctx* c) {
    if (c->comm && R.destroy) R.destroy(c->comm);
    c->comm = nullptr;
    return FEMCY_OK;
}

}  // namespace femcy
