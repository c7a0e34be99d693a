// RCCL behind the C ABI: the exchange step of the row-partitioned product (all-gather of the rows of the
// per-layer dense matrix, or the exchange of the referenced rows only) for callers that bind the library
// without torch.distributed.  RCCL is opened at the first call (dlopen), so the library has no link-time
// dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"

namespace gcn {

// the slice of RCCL's C API used here (rccl.h): handles are opaque, ncclFloat32 = 7, ncclSuccess = 0
typedef void* comm_t;
struct UniqueId {
    char internal[128];
};
struct Rccl {
    void* handle;
    int (*GetUniqueId)(UniqueId*);
    int (*CommInitRank)(comm_t*, int, UniqueId, int);
    int (*CommDestroy)(comm_t);
    int (*CommCount)(comm_t, int*);
    int (*CommUserRank)(comm_t, int*);
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t);
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t);
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*CommAbort)(comm_t);
    int (*GetVersion)(int*);
    const char* (*GetErrorString)(int);
};
constexpr int kFloat32 = 7;

static Rccl g_rccl;
static std::mutex g_rccl_mutex;

static int rccl(const Rccl** out) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.handle) {
        const char* names[] = {getenv("GCN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        void* h = nullptr;
        for (const char* name : names) {
            if (name && *name) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return fail(GCN_ERR_RCCL, "cannot open RCCL (librccl.so): %s", dlerror());
        Rccl r;
        r.handle = h;
#define GCN_SYM(field, symbol)                                                             \
    *(void**)(&r.field) = dlsym(h, symbol);                                                \
    if (!r.field) {                                                                        \
        dlclose(h);                                                                        \
        return fail(GCN_ERR_RCCL, "RCCL has no symbol %s", symbol);                        \
    }
        GCN_SYM(GetUniqueId, "ncclGetUniqueId")
        GCN_SYM(CommInitRank, "ncclCommInitRank")
        GCN_SYM(CommDestroy, "ncclCommDestroy")
        GCN_SYM(CommCount, "ncclCommCount")
        GCN_SYM(CommUserRank, "ncclCommUserRank")
        GCN_SYM(AllGather, "ncclAllGather")
        GCN_SYM(Send, "ncclSend")
        GCN_SYM(Recv, "ncclRecv")
        GCN_SYM(GroupStart, "ncclGroupStart")
        GCN_SYM(GroupEnd, "ncclGroupEnd")
        GCN_SYM(CommAbort, "ncclCommAbort")
        GCN_SYM(GetVersion, "ncclGetVersion")
        GCN_SYM(GetErrorString, "ncclGetErrorString")
#undef GCN_SYM
        // The slice above (ncclFloat32 = 7, the 128-byte id passed by value) is the ABI of NCCL 2.x, which
        // every RCCL release so far follows; a library that reports another major version is refused rather
        // than called with a layout it may not have.  Version code: major * 1000 + ... up to 2.8, major *
        // 10000 + ... from 2.9 on.
        int version = 0;
        if (r.GetVersion(&version) != 0 || version <= 0) {
            dlclose(h);
            return fail(GCN_ERR_RCCL, "ncclGetVersion failed");
        }
        const int major = version >= 10000 ? version / 10000 : version / 1000;
        if (major != 2) {
            dlclose(h);
            return fail(GCN_ERR_RCCL, "RCCL reports version code %d (major %d): only the NCCL 2.x ABI is supported",
                        version, major);
        }
        g_rccl = r;
    }
    *out = &g_rccl;
    return GCN_OK;
}

#define GCN_RCCL_TRY(lib, expr)                                                                          \
    do {                                                                                                 \
        const int r__ = (expr);                                                                          \
        if (r__ != 0) return ::gcn::fail(GCN_ERR_RCCL, "%s failed: %s", #expr, (lib)->GetErrorString(r__)); \
    } while (0)

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_comm_unique_id(void* id128_host) {
    if (!id128_host) return fail(GCN_ERR_BAD_ARG, "id buffer is null");
    const Rccl* lib;
    int rc = rccl(&lib);
    if (rc != GCN_OK) return rc;
    GCN_RCCL_TRY(lib, lib->GetUniqueId((UniqueId*)id128_host));
    return GCN_OK;
}

int gcn_comm_init_rank(void** comm_out, int n_ranks, const void* id128_host, int rank) {
    if (!comm_out || !id128_host || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(GCN_ERR_BAD_ARG, "bad communicator arguments (rank %d of %d)", rank, n_ranks);
    *comm_out = nullptr;
    const Rccl* lib;
    int rc = rccl(&lib);
    if (rc != GCN_OK) return rc;
    UniqueId id;
    memcpy(&id, id128_host, sizeof(id));
    comm_t c = nullptr;
    GCN_RCCL_TRY(lib, lib->CommInitRank(&c, n_ranks, id, rank));
    *comm_out = c;
    return GCN_OK;
}

int gcn_comm_destroy(void* comm) {
    if (!comm) return GCN_OK;
    const Rccl* lib;
    int rc = rccl(&lib);
    if (rc != GCN_OK) return rc;
    GCN_RCCL_TRY(lib, lib->CommDestroy((comm_t)comm));
    return GCN_OK;
}

int gcn_allgather_rows_f32(void* comm, const float* send, float* recv, int64_t rows_per_rank, int64_t f, void* stream) {
    if (!comm || rows_per_rank < 0 || f < 0) return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (rows_per_rank == 0 || f == 0) return GCN_OK;
    if (!send || !recv) return fail(GCN_ERR_BAD_ARG, "null buffer");
    const Rccl* lib;
    int rc = rccl(&lib);
    if (rc != GCN_OK) return rc;
    GCN_RCCL_TRY(lib, lib->AllGather(send, recv, (size_t)rows_per_rank * (size_t)f, kFloat32, (comm_t)comm,
                                     (hipStream_t)stream));
    return GCN_OK;
}

int gcn_exchange_rows_f32(void* comm, const float* send, const int64_t* send_rows_host, float* recv,
                          const int64_t* recv_rows_host, int64_t f, void* stream) {
    if (!comm || !send_rows_host || !recv_rows_host || f < 0) return fail(GCN_ERR_BAD_ARG, "bad argument");
    const Rccl* lib;
    int rc = rccl(&lib);
    if (rc != GCN_OK) return rc;
    int n_ranks = 0;
    GCN_RCCL_TRY(lib, lib->CommCount((comm_t)comm, &n_ranks));
    int64_t total_send = 0, total_recv = 0;
    for (int p = 0; p < n_ranks; ++p) {
        if (send_rows_host[p] < 0 || recv_rows_host[p] < 0) return fail(GCN_ERR_BAD_ARG, "negative row count");
        total_send += send_rows_host[p];
        total_recv += recv_rows_host[p];
    }
    if (f == 0 || (total_send == 0 && total_recv == 0)) return GCN_OK;
    if ((total_send > 0 && !send) || (total_recv > 0 && !recv)) return fail(GCN_ERR_BAD_ARG, "null buffer");
    // one send and one receive per peer, all inside one group: every xGMI link carries its own pair at once
    GCN_RCCL_TRY(lib, lib->GroupStart());
    int64_t s_at = 0, r_at = 0;
    int first_error = 0;
    for (int p = 0; p < n_ranks && !first_error; ++p) {
        if (send_rows_host[p] > 0)
            first_error = lib->Send(send + s_at * f, (size_t)send_rows_host[p] * (size_t)f, kFloat32, p, (comm_t)comm,
                                    (hipStream_t)stream);
        if (!first_error && recv_rows_host[p] > 0)
            first_error = lib->Recv(recv + r_at * f, (size_t)recv_rows_host[p] * (size_t)f, kFloat32, p, (comm_t)comm,
                                    (hipStream_t)stream);
        s_at += send_rows_host[p];
        r_at += recv_rows_host[p];
    }
    const int end = lib->GroupEnd();
    if (first_error || end) {
        // Some of this rank's pairs were never posted (or the group failed): the peers' matching calls
        // would wait for them for ever.  Abort the communicator -- that ends the peers' operations with an
        // error instead of a hang, and frees the communicator -- and report it: after GCN_ERR_RCCL the
        // handle is dead (no further call on it, gcn_comm_destroy included).
        (void)lib->CommAbort((comm_t)comm);
        return fail(GCN_ERR_RCCL, "%s failed: %s; the communicator was aborted and cannot be used again",
                    first_error ? "ncclSend/ncclRecv" : "ncclGroupEnd", lib->GetErrorString(first_error ? first_error : end));
    }
    return GCN_OK;
}

}  // extern "C"
