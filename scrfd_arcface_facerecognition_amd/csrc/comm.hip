// The path's one collective behind the C-ABI: an all-gather of per-rank device buffers over RCCL (xGMI inside a node), so a host
// without torch can run the multi-GPU step (the reference has no analogue: it is single-process).
//   fid_comm_unique_id  : rank 0 makes the 128-byte rendezvous id; the HOST distributes it (file, env, MPI, a torch store ...)
//   fid_comm_init_rank  : one communicator per process / GPU
//   fid_allgather       : ncclAllGather on the context's stream -- ordered with the kernels before and after it, no host sync
// RCCL is resolved at run time (dlopen "librccl.so.1"): the library has no link-time dependency on it, and a process that already
// loaded RCCL (torch ships one with the same SONAME) shares that copy instead of loading a second one.
#include <dlfcn.h>

#include "common.h"

namespace fid {
namespace {

struct UniqueId { char internal[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                        // ncclComm_t
enum { DT_INT8 = 0 };                      // ncclInt8 / ncclChar

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return FID_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("RCCL not found: %s", dlerror());
        return FID_E_STATE;
    }
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        set_error("RCCL library lacks a required symbol");
        return FID_E_STATE;
    }
    g_rccl = r;
    return FID_OK;
}

#define FID_RCCL(call)                                                                                   \
    do {                                                                                                 \
        int r_ = (call);                                                                                 \
        if (r_ != 0) {                                                                                   \
            fid::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, fid::g_rccl.GetErrorString(r_)); \
            return FID_E_HIP;                                                                            \
        }                                                                                                \
    } while (0)

}  // namespace
}  // namespace fid

struct fid_comm {
    fid::Comm comm = nullptr;
    int nranks = 0, rank = 0;
};

extern "C" {

int fid_comm_unique_id(void *id_out, size_t bytes) {
    FID_REQUIRE(id_out && bytes == FID_COMM_ID_BYTES, "id buffer must be FID_COMM_ID_BYTES (%d) bytes", FID_COMM_ID_BYTES);
    FID_TRY(fid::load_rccl());
    fid::UniqueId id;
    FID_RCCL(fid::g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, sizeof(id.internal));
    return FID_OK;
}

int fid_comm_init_rank(fid_ctx *ctx, int nranks, int rank, const void *id, size_t bytes, fid_comm **out) {
    FID_REQUIRE(ctx && id && out && bytes == FID_COMM_ID_BYTES, "bad args");
    FID_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d outside [0, %d)", rank, nranks);
    FID_TRY(fid::load_rccl());
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));          // the communicator binds to the calling thread's current device
    fid::UniqueId uid;
    memcpy(uid.internal, id, sizeof(uid.internal));
    fid_comm *c = new fid_comm();
    int r = fid::g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
    if (r != 0) {
        fid::set_error("ncclCommInitRank(%d of %d) -> %s", rank, nranks, fid::g_rccl.GetErrorString(r));
        delete c;
        return FID_E_HIP;
    }
    c->nranks = nranks;
    c->rank = rank;
    *out = c;
    return FID_OK;
}

int fid_comm_destroy(fid_ctx *ctx, fid_comm *comm) {
    if (!comm) return FID_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (comm->comm && fid::g_rccl.CommDestroy) (void)fid::g_rccl.CommDestroy(comm->comm);
    delete comm;
    return FID_OK;
}

int fid_comm_info(fid_comm *comm, int *nranks, int *rank) {
    FID_REQUIRE(comm, "comm is NULL");
    if (nranks) *nranks = comm->nranks;
    if (rank) *rank = comm->rank;
    return FID_OK;
}

// recv_dev [nranks][bytes_per_rank] <- every rank's send_dev [bytes_per_rank], in rank order; send_dev may be the rank's own
// block of recv_dev (in place).
int fid_allgather(fid_ctx *ctx, fid_comm *comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank) {
    FID_REQUIRE(ctx && comm && comm->comm && send_dev && recv_dev && bytes_per_rank > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_RCCL(fid::g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, fid::DT_INT8, comm->comm, ctx->stream));
    return FID_OK;
}

}  // extern "C"
