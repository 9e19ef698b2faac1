// The ONE collective of the path, owned by the library: the metric gather / reduction at the end of an evaluation over the ranks of
// a node (one process per GPU; tuples are sharded, there is no data-path collective - DESIGN.md 6).
//
// Reference: /root/reference/train.py:270-277 (init_process_group, backend "nccl") and train.py:102-106 (the validation loss's
// 1-element all_reduce - the reference's only explicit collective); eval metrics are per-pair pose errors, gathered for the exact
// sort-based AUC (e2e_multi_view_matching_amd/distributed.py).
//
// RCCL over xGMI.  librccl.so.1 is opened at the FIRST e2emv_comm_* call (dlopen): the library itself links only libamdhip64 and
// libstdc++, a single-GPU host never maps RCCL, and a process that already carries an RCCL (torch's) shares that copy (same
// soname).  Bootstrap: ncclGetUniqueId on rank 0, its 128 bytes handed to the other ranks by whatever the host has - a shared
// file (e2emv_comm_init_file: a plain-C / numpy host needs nothing else), an environment variable, or the launcher's own store
// (Python: torch.distributed.broadcast_object_list).  Latency-bound (a few KB per evaluation): ring bandwidth is irrelevant.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstring>
#include <thread>

#include "common.h"

namespace {

// the slice of rccl.h this file needs (NCCL ABI: stable enum values, opaque handles)
typedef struct ncclComm* rcclComm_t;
struct rcclUniqueId { char internal[E2EMV_COMM_ID_BYTES]; };
enum { RCCL_SUCCESS = 0, RCCL_FLOAT32 = 7, RCCL_SUM = 0, RCCL_MAX = 2, RCCL_MIN = 3 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;  // why it could not be loaded
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.why = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.handle, n); if (!p && api.why.empty()) api.why = std::string("librccl.so.1 lacks ") + n; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return &api;
}

}  // namespace

struct e2emv_comm {
    rcclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

using namespace e2emv;

#define E2EMV_RCCL(ctx, api, expr)                                                                                          \
    do {                                                                                                                    \
        int _r = (expr);                                                                                                    \
        if (_r != RCCL_SUCCESS)                                                                                             \
            return set_err(ctx, E2EMV_EHIP, "%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(_r) : "?"); \
    } while (0)

extern "C" int e2emv_comm_unique_id(e2emv_ctx* ctx, void* id_out) {
    if (!ctx || !id_out) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    RcclApi* api = rccl();
    if (!api->why.empty()) return set_err(ctx, E2EMV_ESTATE, "RCCL is not available: %s", api->why.c_str());
    (void)hipSetDevice(ctx->device);
    rcclUniqueId id;
    E2EMV_RCCL(ctx, api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return E2EMV_OK;
}

extern "C" int e2emv_comm_init(e2emv_ctx* ctx, const void* id, int rank, int world, e2emv_comm** out) {
    if (!ctx || !id || !out) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return set_err(ctx, E2EMV_EINVAL, "comm_init: rank %d of %d", rank, world);
    RcclApi* api = rccl();
    if (!api->why.empty()) return set_err(ctx, E2EMV_ESTATE, "RCCL is not available: %s", api->why.c_str());
    (void)hipSetDevice(ctx->device);
    rcclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    e2emv_comm* c = new (std::nothrow) e2emv_comm();
    if (!c) return set_err(ctx, E2EMV_ENOMEM, "comm_init: out of host memory");
    c->rank = rank; c->world = world; c->device = ctx->device;
    int r = api->CommInitRank(&c->comm, world, uid, rank);  // collective: returns when every rank of `world` has called it
    if (r != RCCL_SUCCESS) {
        delete c;
        return set_err(ctx, E2EMV_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString ? api->GetErrorString(r) : "?");
    }
    *out = c;
    return E2EMV_OK;
}

// Bootstrap through a file every rank can see (one node: /dev/shm or /tmp): rank 0 writes the id to `path`.tmp and renames it (the
// others never see a partial file); the others poll for it.  The file carries the id only - remove it after the job.
extern "C" int e2emv_comm_init_file(e2emv_ctx* ctx, const char* path, int rank, int world, double timeout_s, e2emv_comm** out) {
    if (!ctx || !path || !out) return E2EMV_EINVAL;
    char id[E2EMV_COMM_ID_BYTES];
    if (rank == 0) {
        if (int rc = e2emv_comm_unique_id(ctx, id)) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); E2EMV_LOCK(ctx); return set_err(ctx, E2EMV_EINVAL, "comm_init_file: cannot write %s", tmp.c_str()); }
        fclose(f);
        if (rename(tmp.c_str(), path) != 0) { E2EMV_LOCK(ctx); return set_err(ctx, E2EMV_EINVAL, "comm_init_file: cannot rename %s", tmp.c_str()); }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            struct stat st;
            if (stat(path, &st) == 0 && st.st_size == (off_t)sizeof id) {
                FILE* f = fopen(path, "rb");
                const bool ok = f && fread(id, 1, sizeof id, f) == sizeof id;
                if (f) fclose(f);
                if (ok) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                E2EMV_LOCK(ctx);
                return set_err(ctx, E2EMV_ESTATE, "comm_init_file: rank %d waited %.0f s for %s", rank, timeout_s, path);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    return e2emv_comm_init(ctx, id, rank, world, out);
}

extern "C" int e2emv_comm_destroy(e2emv_ctx* ctx, e2emv_comm* comm) {
    if (!comm) return E2EMV_OK;
    RcclApi* api = rccl();
    if (comm->comm && api->CommDestroy) {
        (void)hipSetDevice(comm->device);
        (void)api->CommDestroy(comm->comm);
    }
    delete comm;
    (void)ctx;
    return E2EMV_OK;
}

extern "C" int e2emv_comm_rank(const e2emv_comm* comm, int* rank, int* world) {
    if (!comm) return E2EMV_EINVAL;
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return E2EMV_OK;
}

// every rank contributes n floats; d_all [world][n] in rank order on every rank.  Ragged contributions: gather the counts first
// (n = 1), pad to the maximum (distributed.py does).  Stream-ordered like every other call.
extern "C" int e2emv_metric_allgather(e2emv_ctx* ctx, e2emv_comm* comm, const float* d_local, int n, float* d_all, void* stream) {
    if (!ctx || !comm || !d_local || !d_all) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (n < 0) return set_err(ctx, E2EMV_ESHAPE, "metric_allgather: n = %d", n);
    if (n == 0) return E2EMV_OK;
    RcclApi* api = rccl();
    E2EMV_RCCL(ctx, api, api->AllGather(d_local, d_all, (size_t)n, RCCL_FLOAT32, comm->comm, (hipStream_t)stream));
    return E2EMV_OK;
}

// in place: d_buf[i] = op over the ranks (train.py:102-106: the validation loss's all_reduce; bench.py: MAX of the step time)
extern "C" int e2emv_metric_allreduce(e2emv_ctx* ctx, e2emv_comm* comm, float* d_buf, int n, int op, void* stream) {
    if (!ctx || !comm || !d_buf) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (n < 0 || op < E2EMV_REDUCE_SUM || op > E2EMV_REDUCE_MIN) return set_err(ctx, E2EMV_EINVAL, "metric_allreduce: n = %d, op = %d", n, op);
    if (n == 0) return E2EMV_OK;
    RcclApi* api = rccl();
    const int rop = op == E2EMV_REDUCE_SUM ? RCCL_SUM : (op == E2EMV_REDUCE_MAX ? RCCL_MAX : RCCL_MIN);
    E2EMV_RCCL(ctx, api, api->AllReduce(d_buf, d_buf, (size_t)n, RCCL_FLOAT32, rop, comm->comm, (hipStream_t)stream));
    return E2EMV_OK;
}
