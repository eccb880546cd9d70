// pvnet_rccl.hip -- the path's ONE exchange, issued by the library itself ("the RCCL C API from the extension on the
// same stream"): an all-gather of the ranks' [B/G, vn, 2] float32 key-points with ncclAllGather on the caller's VOTING stream.
// Replaces the gather of DataParallel(EvalWrapper) (tools/train_linemod.py:183-184, tools/demo.py:174).  Why not torch.distributed's
// collective: its ProcessGroup launches on an internal stream, which takes a fourth hardware queue beside the voting streams -- every
// rank of an N > 1 run then starts 3.5 % below the N = 1 rate (profiles/r05c_gather_stream_ab.txt).  A collective on the stream that
// voted needs no event, no extra stream and no cross-stream wait: stream order is the dependency.
// librccl is loaded at run time (dlopen), never linked: a process that does not gather does not need it.  torch.distributed (any
// backend) stays the bootstrap -- the 128-byte ncclUniqueId is the only thing it has to broadcast (pvnet_amd/distributed.py).
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "pvnet_vote.h"

namespace {

struct UniqueId { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value to ncclCommInitRank
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, void*);
typedef int (*CommCountFn)(void*, int*);
constexpr int kNcclFloat32 = 7;            // ncclDataType_t: ncclFloat32 = ncclFloat = 7

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    CommCountFn comm_count = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

bool load_from(const char* path) {
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    Rccl r;
    r.handle = h;
    r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
    r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(h, "ncclAllGather"));
    r.comm_count = reinterpret_cast<CommCountFn>(dlsym(h, "ncclCommCount"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.comm_count) {
        dlclose(h);
        return false;
    }
    g_rccl = r;
    return true;
}

}  // namespace

extern "C" {

int pvnet_rccl_load(const char* path) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_rccl.handle) return 0;
    if (path && *path) return load_from(path) ? 0 : PVNET_E_UNSUPPORTED;
    for (const char* p : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
        if (load_from(p)) return 0;
    return PVNET_E_UNSUPPORTED;
}

int pvnet_rccl_unique_id(void* id128) {
    if (!id128) return PVNET_E_BADARG;
    if (!g_rccl.handle && pvnet_rccl_load(nullptr)) return PVNET_E_UNSUPPORTED;
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc == 0) memcpy(id128, &id, sizeof(id));
    return rc;   // ncclResult_t: 0 = ncclSuccess
}

int pvnet_rccl_comm_init(void** comm, int nranks, const void* id128, int rank) {
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return PVNET_E_BADARG;
    if (!g_rccl.handle && pvnet_rccl_load(nullptr)) return PVNET_E_UNSUPPORTED;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    return g_rccl.comm_init_rank(comm, nranks, id, rank);   // on the CURRENT device; collective over the ranks
}

int pvnet_rccl_comm_destroy(void* comm) {
    if (!comm) return PVNET_E_BADARG;
    if (!g_rccl.handle) return PVNET_E_UNSUPPORTED;
    return g_rccl.comm_destroy(comm);
}

int pvnet_rccl_comm_ranks(void* comm, int* nranks) {
    if (!comm || !nranks) return PVNET_E_BADARG;
    if (!g_rccl.handle) return PVNET_E_UNSUPPORTED;
    return g_rccl.comm_count(comm, nranks);
}

int pvnet_vote_allgather(const float* local, float* all, size_t count_per_rank, void* comm, void* stream) {
    if (!local || !all || !comm) return PVNET_E_BADARG;
    if (!g_rccl.handle) return PVNET_E_UNSUPPORTED;
    if (count_per_rank == 0) return 0;
    return g_rccl.all_gather(local, all, count_per_rank, kNcclFloat32, comm, stream);   // no sync, no allocation: one launch on `stream`
}

}  // extern "C"
