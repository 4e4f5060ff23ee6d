// kws_comm.cpp -- the one collective of the path: clips shard contiguously over the GPUs of a node, nothing is
// exchanged inside the pipeline, and the per-clip scores [B][C] of every rank are all-gathered over xGMI with RCCL.  The host
// side stays C: RCCL is reached through its own C API (ncclGetUniqueId / ncclCommInitRank / ncclAllGather), resolved with
// dlopen so that a single-GPU application does not need librccl at all.  The 128-byte unique id is created on rank 0 and
// carried to the other ranks by whatever the application already uses to start its processes (bench.py: torch.distributed).
#include "kws_internal.h"

#include <dlfcn.h>

namespace {
typedef int ncclResult;
struct NcclId { char internal[128]; };
struct Rccl {
    void *lib = nullptr;
    ncclResult (*GetUniqueId)(NcclId *) = nullptr;
    ncclResult (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    ncclResult (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    ncclResult (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(ncclResult) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

EI_IMPULSE_ERROR load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return EI_IMPULSE_OK;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;      // a copy the process already holds (PyTorch's) is reused
    if (!lib) return fail(KWS_ERROR_HIP, "librccl not found: %s", dlerror());
    Rccl r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
    r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
        dlclose(lib);
        return fail(KWS_ERROR_HIP, "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    }
    r.lib = lib;
    g_rccl = r;
    return EI_IMPULSE_OK;
}
}  // namespace

struct kws_comm {
    void *comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

#pragma GCC visibility push(default)
extern "C" {

EI_IMPULSE_ERROR kws_comm_unique_id(void *id, size_t nbytes)
{
    if (!id || nbytes < KWS_COMM_ID_BYTES) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_comm_unique_id needs a %d-byte buffer", KWS_COMM_ID_BYTES);
    EI_IMPULSE_ERROR e = load_rccl();
    if (e) return e;
    NcclId u;
    const ncclResult r = g_rccl.GetUniqueId(&u);
    if (r != 0) return fail(KWS_ERROR_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    memcpy(id, u.internal, KWS_COMM_ID_BYTES);
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_comm_create(const void *id, size_t nbytes, int world_size, int rank, int device, kws_comm **out)
{
    if (!id || !out || nbytes < KWS_COMM_ID_BYTES || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(KWS_ERROR_BAD_ARGUMENT, "kws_comm_create: bad argument");
    *out = nullptr;
    EI_IMPULSE_ERROR e = load_rccl();
    if (e) return e;
    HIP_TRY(hipSetDevice(device));
    NcclId u;
    memcpy(u.internal, id, KWS_COMM_ID_BYTES);
    kws_comm *c = new kws_comm();
    c->world = world_size; c->rank = rank; c->device = device;
    const ncclResult r = g_rccl.CommInitRank(&c->comm, world_size, u, rank);
    if (r != 0) { delete c; return fail(KWS_ERROR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world_size, g_rccl.GetErrorString(r)); }
    *out = c;
    return EI_IMPULSE_OK;
}

int kws_comm_world_size(const kws_comm *c) { return c ? c->world : 0; }
int kws_comm_rank(const kws_comm *c) { return c ? c->rank : -1; }

EI_IMPULSE_ERROR kws_allgather_scores(kws_comm *c, const float *local_scores, float *all_scores, size_t clips_per_rank, int label_count,
                                      void *stream)
{
    if (!c || !local_scores || !all_scores || label_count < 1) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_allgather_scores: bad argument");
    if (clips_per_rank == 0) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(c->device));
    const ncclResult r = g_rccl.AllGather(local_scores, all_scores, clips_per_rank * (size_t)label_count, 7 /* ncclFloat32 */, c->comm,
                                          (hipStream_t)stream);
    if (r != 0) return fail(KWS_ERROR_HIP, "ncclAllGather: %s", g_rccl.GetErrorString(r));
    return EI_IMPULSE_OK;
}

void kws_comm_destroy(kws_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
#pragma GCC visibility pop
