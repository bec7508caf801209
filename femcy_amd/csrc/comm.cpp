// RCCL plumbing for the element-partitioned solve.  The reference is single-device (SURVEY.md 2:
// "Collectives: none"); this is new work.  One process per GPU; each rank holds a sub-mesh whose
// sub-assembled K has partial sums on interface rows.  Per CG iteration there are two collectives:
//   1. all-reduce(sum, f64) of the packed global interface vector of y = K_loc d with the local
//      d.K_loc.d appended (d^T A d = sum_ranks d_r^T K_r d_r, so no owner mask is needed there),
//   2. all-gather of the per-rank (r.M.r over owned DOFs, max|r|) pair.
// RCCL is bound with dlopen so that libfemcy_hip.so shares whatever librccl the host process already
// loaded (PyTorch bundles its own copy with the same SONAME) and single-GPU users need none.
//
// A second transport, the in-process group, joins several contexts of ONE process (one host thread each, any
// devices) through host staging buffers and a condition-variable rendezvous.  It exists so that the whole
// multi-rank device path (sub-assembled K, interface pack / unpack, owner masks, the two collectives per
// iteration) can be run and checked on a single GPU; ranks sum in rank order, so every rank gets the same bits.
//
// A third transport, the shared-memory group (round 4), does the same between PROCESSES of one host through a POSIX
// shared-memory segment (staging areas + a generation barrier on lock-free atomics).  It needs no RCCL, so N processes
// can share ONE GPU -- which RCCL refuses -- and that is what lets the cross-process branches of the mailbox path
// (hipIpcGetMemHandle / hipIpcOpenMemHandle of a fine-grained allocation, kernels of different processes polling each
// other's words) run on a one-GPU box before they meet a real multi-GPU node.
#include <dlfcn.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include "ctx.hpp"
#include "shm_group.hpp"

namespace femcy {

typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(void**, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_sendrecv)(void*, size_t, int, int, void*, hipStream_t);   // ncclSend (const void*) / ncclRecv
typedef int (*fn_group)(void);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

static struct {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_allgather allgather = nullptr;
    fn_sendrecv send = nullptr, recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
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
    R.send = (fn_sendrecv)dlsym(R.lib, "ncclSend");
    R.recv = (fn_sendrecv)dlsym(R.lib, "ncclRecv");
    R.group_start = (fn_group)dlsym(R.lib, "ncclGroupStart");
    R.group_end = (fn_group)dlsym(R.lib, "ncclGroupEnd");
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

// ------------------------------------------------------------------------------ in-process group
static const char LOCAL_MAGIC[8] = {'F', 'E', 'M', 'C', 'Y', 'L', 'O', 'C'};

struct LocalGroup {
    std::mutex m;
    std::condition_variable cv;
    int nranks = 0, arrived = 0, joined = 0, left = 0;
    uint64_t gen = 0;
    bool broken = false;
    std::vector<std::vector<double>> stage;   // one buffer per rank
    std::vector<std::vector<int32_t>> nb_rank, nb_ptr;   // per rank: its neighbour segment table (neighbour exchange)
};
static std::mutex g_groups_m;
static std::map<uint64_t, std::shared_ptr<LocalGroup>> g_groups;
static uint64_t g_next_token = 1;

int comm_local_id(void* id128) {
    std::memset(id128, 0, 128);
    std::memcpy(id128, LOCAL_MAGIC, 8);
    std::lock_guard<std::mutex> lk(g_groups_m);
    const uint64_t token = g_next_token++;
    std::memcpy((char*)id128 + 8, &token, 8);
    return FEMCY_OK;
}

// all ranks arrive or the call fails after 60 s (a rank that died must not hang the others)
static int local_barrier(LocalGroup* g) {
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) {
        set_error("in-process group: a previous rendezvous failed");
        return FEMCY_ECOMM;
    }
    const uint64_t gen0 = g->gen;
    if (++g->arrived == g->nranks) {
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
        return FEMCY_OK;
    }
    if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->gen != gen0 || g->broken; }) || g->broken) {
        g->broken = true;
        g->cv.notify_all();
        set_error("in-process group: rendezvous timed out (%d of %d ranks arrived)", g->arrived, g->nranks);
        return FEMCY_ECOMM;
    }
    return FEMCY_OK;
}

static int local_init(Ctx* c, int32_t rank, int32_t nranks, const void* id128) {
    uint64_t token;
    std::memcpy(&token, (const char*)id128 + 8, 8);
    std::shared_ptr<LocalGroup> g;
    {
        std::lock_guard<std::mutex> lk(g_groups_m);
        auto& slot = g_groups[token];
        if (!slot) {
            slot = std::make_shared<LocalGroup>();
            slot->nranks = nranks;
            slot->stage.resize(nranks);
            slot->nb_rank.resize(nranks);
            slot->nb_ptr.resize(nranks);
        }
        g = slot;
    }
    {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->nranks != nranks || g->joined >= nranks) {
            set_error("in-process group: rank %d of %d does not fit the group (%d ranks, %d joined)", rank, nranks,
                      g->nranks, g->joined);
            return FEMCY_ECOMM;
        }
        ++g->joined;
    }
    c->comm = g.get();
    c->comm_kind = COMM_LOCAL;
    c->comm_local = true;
    c->comm_token = token;
    c->rank = rank;
    c->nranks = nranks;
    return FEMCY_OK;
}

static int local_exchange(Ctx* c, const double* d_send, int64_t count, double* d_recv, bool gather) {
    LocalGroup* g = (LocalGroup*)c->comm;
    std::vector<double>& mine = g->stage[c->rank];
    mine.resize((size_t)count);
    FEMCY_HIP(hipMemcpyAsync(mine.data(), d_send, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    int rc = local_barrier(g);
    if (rc) return rc;
    std::vector<double> out((size_t)(gather ? count * g->nranks : count), 0.0);
    for (int r = 0; r < g->nranks; ++r) {       // rank order: the same bits on every rank
        if ((int64_t)g->stage[r].size() != count) {
            set_error("in-process group: rank %d sent %zu values, rank %d sent %lld", r, g->stage[r].size(), c->rank,
                      (long long)count);
            return FEMCY_ECOMM;
        }
        if (gather)
            std::memcpy(out.data() + (size_t)r * count, g->stage[r].data(), sizeof(double) * count);
        else
            for (int64_t i = 0; i < count; ++i) out[i] += g->stage[r][i];
    }
    if ((rc = local_barrier(g))) return rc;     // everyone has read every staging buffer
    FEMCY_HIP(hipMemcpyAsync(d_recv, out.data(), sizeof(double) * out.size(), hipMemcpyHostToDevice, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    return FEMCY_OK;
}

// ------------------------------------------------------------------------------ shared-memory group (processes)
// shm_group.hpp: rendezvous, staging areas, sums in rank order; here only the device copies around it
int comm_shm_id(void* id128, int64_t cap_doubles) {
    shm_make_id(id128, cap_doubles);
    return FEMCY_OK;
}

static int shm_init(Ctx* c, int32_t rank, int32_t nranks, const void* id128) {
    FEMCY_REQUIRE(nranks <= SHM_MAXR, "the shared-memory group supports up to %d ranks", SHM_MAXR);
    auto g = std::make_unique<ShmGroup>();
    if (!g->open(rank, nranks, id128)) {
        set_error("%s", g->err.c_str());
        return FEMCY_ECOMM;
    }
    c->comm = g.release();
    c->comm_kind = COMM_SHM;
    c->comm_local = false;
    c->rank = rank;
    c->nranks = nranks;
    return FEMCY_OK;
}

static int shm_exchange(Ctx* c, const double* d_send, int64_t count, double* d_recv, bool gather) {
    ShmGroup* g = (ShmGroup*)c->comm;
    FEMCY_REQUIRE((uint64_t)count <= g->h->cap, "shared-memory group: %lld values exceed the staging area of %llu (femcy_comm_shm_id)",
                  (long long)count, (unsigned long long)g->h->cap);
    FEMCY_HIP(hipMemcpyAsync(g->area(c->rank), d_send, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> out((size_t)(gather ? count * c->nranks : count));
    if (!g->exchange(c->rank, g->area(c->rank), count, out.data(), gather)) {
        set_error("%s", g->err.c_str());
        return FEMCY_ECOMM;
    }
    FEMCY_HIP(hipMemcpyAsync(d_recv, out.data(), sizeof(double) * out.size(), hipMemcpyHostToDevice, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    return FEMCY_OK;
}

int comm_unique_id(void* id128) {
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid id;
    FEMCY_NCCL(R.get_uid(&id));
    std::memcpy(id128, &id, sizeof(id));
    return FEMCY_OK;
}

int comm_init(Ctx* c, int32_t rank, int32_t nranks, const void* id128) {
    if (std::memcmp(id128, LOCAL_MAGIC, 8) == 0) return local_init(c, rank, nranks, id128);
    if (std::memcmp(id128, SHM_MAGIC, 8) == 0) return shm_init(c, rank, nranks, id128);
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid id;
    std::memcpy(&id, id128, sizeof(id));
    FEMCY_HIP(hipSetDevice(c->device));
    FEMCY_NCCL(R.init_rank(&c->comm, nranks, id, rank));
    c->comm_kind = COMM_RCCL;
    c->rank = rank;
    c->nranks = nranks;
    return FEMCY_OK;
}

int comm_allreduce_sum(Ctx* c, double* d_buf, int64_t count) {
    if (!c->comm || count <= 0) return FEMCY_OK;
    if (c->comm_local) return local_exchange(c, d_buf, count, d_buf, false);
    if (c->comm_kind == COMM_SHM) return shm_exchange(c, d_buf, count, d_buf, false);
    FEMCY_NCCL(R.allreduce(d_buf, d_buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->comm, c->stream));
    return FEMCY_OK;
}

int comm_allgather(Ctx* c, const double* d_send, double* d_recv, int64_t count) {
    if (!c->comm) {
        FEMCY_HIP(hipMemcpyAsync(d_recv, d_send, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return FEMCY_OK;
    }
    if (c->comm_local) return local_exchange(c, d_send, count, d_recv, true);
    if (c->comm_kind == COMM_SHM) return shm_exchange(c, d_send, count, d_recv, true);
    FEMCY_NCCL(R.allgather(d_send, d_recv, (size_t)count, /*ncclFloat64*/ 8, c->comm, c->stream));
    return FEMCY_OK;
}

int comm_register_neighbours(Ctx* c) {
    if (c->comm && c->comm_kind == COMM_SHM) {
        ShmGroup* g = (ShmGroup*)c->comm;
        const int nnb = (int)c->h_nb_rank.size();
        FEMCY_REQUIRE(nnb <= SHM_MAXR, "the shared-memory group holds up to %d neighbours per rank", SHM_MAXR);
        ShmHeader::Rank& me = g->h->rk[c->rank];
        me.nnb = nnb;
        for (int k = 0; k < nnb; ++k) me.nb_rank[k] = c->h_nb_rank[k];
        for (int k = 0; k <= nnb; ++k) me.nb_ptr[k] = c->h_nb_ptr[k];
        return FEMCY_OK;                                 // read by the peers after the first barrier of an exchange
    }
    if (!c->comm || !c->comm_local) return FEMCY_OK;
    LocalGroup* g = (LocalGroup*)c->comm;
    std::lock_guard<std::mutex> lk(g->m);
    g->nb_rank[c->rank] = c->h_nb_rank;
    g->nb_ptr[c->rank] = c->h_nb_ptr;
    return FEMCY_OK;
}

// every rank sends segment k of d_nb_send to neighbour h_nb_rank[k] and receives that neighbour's segment for it
// into segment k of d_nb_recv (the two sides list the shared DOFs in the same order)
int comm_neighbour_exchange(Ctx* c, hipStream_t stream) {
    if (!c->comm) return FEMCY_OK;
    const int nnb = (int)c->h_nb_rank.size();
    if (c->comm_local) {
        LocalGroup* g = (LocalGroup*)c->comm;
        const int64_t total = c->h_nb_ptr.empty() ? 0 : c->h_nb_ptr.back();
        std::vector<double>& mine = g->stage[c->rank];
        mine.resize((size_t)total);
        if (total) FEMCY_HIP(hipMemcpyAsync(mine.data(), c->d_nb_send, sizeof(double) * total, hipMemcpyDeviceToHost, stream));
        FEMCY_HIP(hipStreamSynchronize(stream));
        int rc = local_barrier(g);
        if (rc) return rc;
        std::vector<double> in((size_t)total);
        for (int k = 0; k < nnb; ++k) {
            const int q = c->h_nb_rank[k];
            const int32_t cnt = c->h_nb_ptr[k + 1] - c->h_nb_ptr[k];
            const std::vector<int32_t>& qr = g->nb_rank[q];
            int kq = -1;
            for (size_t t = 0; t < qr.size(); ++t)
                if (qr[t] == c->rank) kq = (int)t;
            if (kq < 0 || g->nb_ptr[q][kq + 1] - g->nb_ptr[q][kq] != cnt) {
                set_error("in-process group: ranks %d and %d disagree on their shared DOFs", c->rank, q);
                return FEMCY_ECOMM;
            }
            std::memcpy(in.data() + c->h_nb_ptr[k], g->stage[q].data() + g->nb_ptr[q][kq], sizeof(double) * cnt);
        }
        if ((rc = local_barrier(g))) return rc;
        if (total) FEMCY_HIP(hipMemcpyAsync(c->d_nb_recv, in.data(), sizeof(double) * total, hipMemcpyHostToDevice, stream));
        FEMCY_HIP(hipStreamSynchronize(stream));
        return FEMCY_OK;
    }
    if (c->comm_kind == COMM_SHM) {
        ShmGroup* g = (ShmGroup*)c->comm;
        ShmHeader* h = g->h;
        const int64_t total = c->h_nb_ptr.empty() ? 0 : c->h_nb_ptr.back();
        FEMCY_REQUIRE((uint64_t)total <= h->cap, "shared-memory group: %lld interface values exceed the staging area", (long long)total);
        if (total) FEMCY_HIP(hipMemcpyAsync(g->area(c->rank), c->d_nb_send, sizeof(double) * total, hipMemcpyDeviceToHost, stream));
        FEMCY_HIP(hipStreamSynchronize(stream));
        if (!g->barrier()) {
            set_error("%s", g->err.c_str());
            return FEMCY_ECOMM;
        }
        std::vector<double> in((size_t)total);
        for (int k = 0; k < nnb; ++k) {
            const int q = c->h_nb_rank[k];
            const int32_t cnt = c->h_nb_ptr[k + 1] - c->h_nb_ptr[k];
            const ShmHeader::Rank& rq = h->rk[q];
            int kq = -1;
            for (int t = 0; t < rq.nnb; ++t)
                if (rq.nb_rank[t] == c->rank) kq = t;
            if (kq < 0 || rq.nb_ptr[kq + 1] - rq.nb_ptr[kq] != cnt) {
                set_error("shared-memory group: ranks %d and %d disagree on their shared DOFs", c->rank, q);
                return FEMCY_ECOMM;
            }
            std::memcpy(in.data() + c->h_nb_ptr[k], g->area(q) + rq.nb_ptr[kq], sizeof(double) * cnt);
        }
        if (!g->barrier()) {
            set_error("%s", g->err.c_str());
            return FEMCY_ECOMM;
        }
        if (total) FEMCY_HIP(hipMemcpyAsync(c->d_nb_recv, in.data(), sizeof(double) * total, hipMemcpyHostToDevice, stream));
        FEMCY_HIP(hipStreamSynchronize(stream));
        return FEMCY_OK;
    }
    if (!R.send || !R.recv || !R.group_start || !R.group_end) {
        set_error("librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
        return FEMCY_ECOMM;
    }
    FEMCY_NCCL(R.group_start());
    for (int k = 0; k < nnb; ++k) {
        const size_t cnt = (size_t)(c->h_nb_ptr[k + 1] - c->h_nb_ptr[k]);
        FEMCY_NCCL(R.send(c->d_nb_send + c->h_nb_ptr[k], cnt, /*ncclFloat64*/ 8, c->h_nb_rank[k], c->comm, stream));
        FEMCY_NCCL(R.recv(c->d_nb_recv + c->h_nb_ptr[k], cnt, /*ncclFloat64*/ 8, c->h_nb_rank[k], c->comm, stream));
    }
    FEMCY_NCCL(R.group_end());
    return FEMCY_OK;
}

// `bytes` host bytes per rank -> recv[nranks][bytes] on every rank, through whatever transport the context has: the
// mailbox blobs travel this way, so a host program needs no second communication library for them
int comm_allgather_host(Ctx* c, const void* send, int32_t bytes, void* recv) {
    FEMCY_REQUIRE(send && recv && bytes > 0 && bytes <= (1 << 20), "allgather_host: 1 .. 2^20 bytes per rank");
    if (!c->comm) {
        std::memcpy(recv, send, (size_t)bytes);
        return FEMCY_OK;
    }
    const int R_ = c->nranks;
    const int64_t nd = (bytes + 7) / 8;                  // carried as doubles (bit patterns are only copied)
    double *d_s = nullptr, *d_r = nullptr;
    FEMCY_HIP(dmalloc(&d_s, sizeof(double) * nd));
    if (dmalloc(&d_r, sizeof(double) * nd * R_) != hipSuccess) {
        (void)hipFree(d_s);
        set_error("allgather_host: hipMalloc failed");
        return FEMCY_ENOMEM;
    }
    std::vector<double> hs((size_t)nd, 0.0), hr((size_t)nd * R_);
    std::memcpy(hs.data(), send, (size_t)bytes);
    int rc = FEMCY_OK;
    if (hipMemcpy(d_s, hs.data(), sizeof(double) * nd, hipMemcpyHostToDevice) != hipSuccess) rc = FEMCY_EHIP;
    if (!rc) rc = comm_allgather(c, d_s, d_r, nd);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = FEMCY_EHIP;
    if (!rc && hipMemcpy(hr.data(), d_r, sizeof(double) * nd * R_, hipMemcpyDeviceToHost) != hipSuccess) rc = FEMCY_EHIP;
    (void)hipFree(d_s);
    (void)hipFree(d_r);
    if (rc == FEMCY_EHIP) set_error("allgather_host: a HIP copy failed");
    if (rc) return rc;
    for (int r = 0; r < R_; ++r) std::memcpy((char*)recv + (size_t)r * bytes, hr.data() + (size_t)r * nd, (size_t)bytes);
    return FEMCY_OK;
}

// ------------------------------------------------------------------------------- mailboxes of the persistent PCG
// (protocol: kernels_pcg_persist.hip).  A mailbox is one buffer of 8-byte words in the rank's own HBM:
//   [2][R] entries d.Ad | [2][R][2] entries (r.M.r, max|r|) | [2][nb_total] entries interface rows of Ad
// (entry = 2 words).  It is exported as a 256-byte blob (IPC handle, process id, device pointer, the neighbour segment
// table) that the host program hands to the other ranks (torch.distributed all-gather in bench.py / distributed.py;
// a plain list for the in-process group).
namespace {
struct MboxBlob {
    char magic[8];
    int32_t rank, nranks;
    int64_t pid;
    uint64_t devptr;
    int32_t device, nnb, nb_total, has_ipc;
    hipIpcMemHandle_t handle;          // 64 bytes
    int32_t nb_rank[8];
    int32_t nb_ptr[9];
    int32_t finegrained;
    uint64_t nonce;                    // of the exporting process: equal pids in different pid namespaces are not one process
    char pad[256 - (8 + 8 + 8 + 8 + 16 + 64 + 32 + 36 + 4 + 8)];
};
static_assert(sizeof(MboxBlob) == 256, "mailbox blob is 256 bytes");
const char MBOX_MAGIC[8] = {'F', 'E', 'M', 'C', 'Y', 'M', 'B', '1'};
}  // namespace

int comm_mailbox_export(Ctx* c, void* blob256) {
    FEMCY_REQUIRE(c->comm, "femcy_comm_init must come first");
    FEMCY_REQUIRE(!c->h_nb_ptr.empty() && c->d_if_ptr, "femcy_comm_set_neighbours must come first");
    const int R = c->nranks;
    FEMCY_REQUIRE(R <= 16, "the mailbox path supports up to 16 ranks");
    const int64_t nb_total = c->h_nb_ptr.back();
    const int64_t words = 12 * (int64_t)R + 4 * nb_total + 16;
    if (!c->d_mbox || c->mbox_words < words) {
        if (c->d_mbox) (void)hipFree(c->d_mbox);
        c->d_mbox = nullptr;
        // fine-grained: remote (xGMI) writes must become visible to this device's polls without a kernel boundary
        c->mbox_finegrained = hipExtMallocWithFlags((void**)&c->d_mbox, sizeof(unsigned long long) * words,
                                                    hipDeviceMallocFinegrained) == hipSuccess;
        if (!c->mbox_finegrained) {
            (void)hipGetLastError();
            FEMCY_HIP(dmalloc(&c->d_mbox, sizeof(unsigned long long) * words));
        }
        c->mbox_words = words;
        FEMCY_HIP(dfill_sync(c->d_mbox, 0, sizeof(unsigned long long) * words));
    }
    MboxBlob b;
    std::memset(&b, 0, sizeof(b));
    std::memcpy(b.magic, MBOX_MAGIC, 8);
    b.rank = c->rank;
    b.nranks = R;
    b.pid = (int64_t)getpid();
    b.nonce = process_nonce();
    b.finegrained = c->mbox_finegrained ? 1 : 0;
    b.devptr = (uint64_t)(uintptr_t)c->d_mbox;
    b.device = c->device;
    b.nb_total = (int32_t)nb_total;
    const int nnb = (int)c->h_nb_rank.size();
    b.nnb = nnb <= 8 ? nnb : -1;                                  // more neighbours than the blob holds: not eligible
    for (int k = 0; k < nnb && k < 8; ++k) b.nb_rank[k] = c->h_nb_rank[k];
    for (int k = 0; k <= nnb && k <= 8; ++k) b.nb_ptr[k] = c->h_nb_ptr[k];
    b.has_ipc = hipIpcGetMemHandle(&b.handle, c->d_mbox) == hipSuccess ? 1 : 0;
    if (!b.has_ipc) (void)hipGetLastError();
    std::memcpy(blob256, &b, sizeof(b));
    return FEMCY_OK;
}

int comm_mailbox_import(Ctx* c, int32_t nblobs, const void* blobs) {
    FEMCY_REQUIRE(c->comm && c->d_mbox, "femcy_comm_mailbox_export must come first");
    FEMCY_REQUIRE(nblobs == c->nranks && blobs, "one blob per rank is needed (%d given, %d ranks)", nblobs, c->nranks);
    const MboxBlob* B = reinterpret_cast<const MboxBlob*>(blobs);
    const int R = c->nranks;
    c->persist_multi_local = c->persist_multi = false;
    for (void* q : c->ipc_opened) (void)hipIpcCloseMemHandle(q);
    c->ipc_opened.clear();
    c->h_peer_mbox.assign((size_t)R, nullptr);
    // remote writes into a coarse-grained allocation are not guaranteed to become visible to a running kernel's polls:
    // without a fine-grained mailbox (here or on a peer) this rank votes for the RCCL loop instead of stalling in the
    // bounded spin of its first solve
    for (int r = 0; r < R; ++r)                                   // nothing of a blob is believed before it is validated
        FEMCY_REQUIRE(std::memcmp(B[r].magic, MBOX_MAGIC, 8) == 0 && B[r].rank == r && B[r].nranks == R,
                      "blob %d is not the mailbox of rank %d of %d", r, r, R);
    bool ok = c->mbox_finegrained;
    for (int r = 0; r < R; ++r) ok = ok && (R == 1 || B[r].finegrained != 0);
    for (int r = 0; r < R; ++r) {
        if (r == c->rank) {
            c->h_peer_mbox[r] = c->d_mbox;
            continue;
        }
        if (B[r].pid == (int64_t)getpid() && B[r].nonce == process_nonce()) {   // same process (in-process group): the pointer itself
            if (B[r].device != c->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(B[r].device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                (void)hipGetLastError();
            }
            c->h_peer_mbox[r] = reinterpret_cast<unsigned long long*>((uintptr_t)B[r].devptr);
        } else if (B[r].has_ipc) {
            void* q = nullptr;
            if (hipIpcOpenMemHandle(&q, B[r].handle, hipIpcMemLazyEnablePeerAccess) == hipSuccess && q) {
                c->ipc_opened.push_back(q);
                c->h_peer_mbox[r] = reinterpret_cast<unsigned long long*>(q);
            } else {
                (void)hipGetLastError();
                ok = false;
            }
        } else {
            ok = false;
        }
    }
    // per-position interface table: {entry in the neighbour's Ad area, entry in mine, neighbour | lower << 8, its nb_total}
    const int nnb = (int)c->h_nb_rank.size();
    std::vector<int32_t> tab((size_t)c->nslices * SLICE * 4, -1);
    if (nnb > 8 || (int64_t)c->h_nb_dofs.size() != (int64_t)c->h_nb_ptr.back()) ok = false;
    for (int k = 0; k < nnb && ok; ++k) {
        const int q = c->h_nb_rank[k];
        const MboxBlob& bq = B[q];
        int kq = -1;
        for (int t = 0; t < bq.nnb; ++t)
            if (bq.nb_rank[t] == c->rank) kq = t;
        const int32_t cnt = c->h_nb_ptr[k + 1] - c->h_nb_ptr[k];
        if (kq < 0 || bq.nb_ptr[kq + 1] - bq.nb_ptr[kq] != cnt || cnt % c->dm != 0) {
            ok = false;
            break;
        }
        for (int32_t j = c->h_nb_ptr[k]; j < c->h_nb_ptr[k + 1]; j += c->dm) {
            const int32_t d0 = c->h_nb_dofs[j];
            bool whole = d0 % c->dm == 0;                         // the node's dm DOFs, consecutive entries
            for (int cc = 1; cc < c->dm && whole; ++cc) whole = c->h_nb_dofs[j + cc] == d0 + cc;
            if (!whole) {
                ok = false;
                break;
            }
            const int64_t p = c->h_pos[d0 / c->dm];
            if (tab[p * 4] >= 0) {                                // a node shared with two neighbours: not this path
                ok = false;
                break;
            }
            tab[p * 4 + 0] = bq.nb_ptr[kq] + (j - c->h_nb_ptr[k]);
            tab[p * 4 + 1] = j;
            tab[p * 4 + 2] = q | (q < c->rank ? 0x100 : 0);
            tab[p * 4 + 3] = bq.nb_total;
        }
    }
    if (c->d_mr_tab) (void)hipFree(c->d_mr_tab);
    c->d_mr_tab = nullptr;
    FEMCY_HIP(dmalloc(&c->d_mr_tab, tab.size() * sizeof(int32_t)));
    FEMCY_HIP(hipMemcpy(c->d_mr_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (c->d_peer_tab) (void)hipFree(c->d_peer_tab);
    c->d_peer_tab = nullptr;
    FEMCY_HIP(dmalloc(&c->d_peer_tab, sizeof(unsigned long long*) * 16));
    FEMCY_HIP(hipMemcpy(c->d_peer_tab, c->h_peer_mbox.data(), sizeof(unsigned long long*) * R, hipMemcpyHostToDevice));
    c->persist_multi_local = ok;
    return FEMCY_OK;
}

// collective: the path is taken only if EVERY rank can take it (mailboxes mapped, one sharer per interface node, the
// pattern fits the persistent kernel) -- a rank that went the other way would leave the others polling
int comm_persist_agree(Ctx* c, int32_t* enabled) {
    FEMCY_REQUIRE(c->comm, "femcy_comm_init must come first");
    const bool local = c->opt_persist_multi && c->persist_multi_local && c->d_mr_tab && persist_pattern_fits(c);
    double flag = local ? 0.0 : 1.0;
    FEMCY_HIP(hipMemcpyAsync(c->d_commbuf, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = comm_allreduce_sum(c, c->d_commbuf, 1);
    if (rc) return rc;
    FEMCY_HIP(hipMemcpyAsync(&flag, c->d_commbuf, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    c->persist_multi = flag == 0.0;
    c->persist_multi_failed = false;
    if (enabled) *enabled = c->persist_multi ? 1 : 0;
    return FEMCY_OK;
}

int comm_destroy(Ctx* c) {
    for (void* q : c->ipc_opened) (void)hipIpcCloseMemHandle(q);
    c->ipc_opened.clear();
    if (c->d_mbox) (void)hipFree(c->d_mbox);
    if (c->d_mr_tab) (void)hipFree(c->d_mr_tab);
    if (c->d_peer_tab) (void)hipFree(c->d_peer_tab);
    c->d_mbox = nullptr;
    c->d_mr_tab = nullptr;
    c->d_peer_tab = nullptr;
    c->persist_multi = c->persist_multi_local = false;
    if (c->comm && c->comm_local) {
        LocalGroup* g = (LocalGroup*)c->comm;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->m);
            last = ++g->left == g->joined;
        }
        if (last) {
            std::lock_guard<std::mutex> lk(g_groups_m);
            g_groups.erase(c->comm_token);
        }
        c->comm = nullptr;
        c->comm_local = false;
        c->comm_kind = COMM_NONE;
        return FEMCY_OK;
    }
    if (c->comm && c->comm_kind == COMM_SHM) {
        ShmGroup* g = (ShmGroup*)c->comm;
        g->leave();
        delete g;
        c->comm = nullptr;
        c->comm_kind = COMM_NONE;
        return FEMCY_OK;
    }
    if (c->comm && R.destroy) R.destroy(c->comm);
    c->comm = nullptr;
    c->comm_kind = COMM_NONE;
    return FEMCY_OK;
}

}  // namespace femcy
