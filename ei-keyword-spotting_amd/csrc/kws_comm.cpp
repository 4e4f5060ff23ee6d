// kws_comm.cpp -- the one collective of the path: clips shard contiguously over the GPUs of a node, nothing is
// exchanged inside the pipeline, and the per-clip scores [B][C] of every rank are all-gathered over xGMI with RCCL.  The host
// side stays C: RCCL is reached through its own C API.  Types, enums and the version this file was built for come from
// <rccl/rccl.h>; the library itself is resolved with dlopen so that a single-GPU application does not need librccl at all.
// The 128-byte unique id is created on rank 0 and carried to the other ranks by whatever the application already uses to start its
// processes (bench.py: torch.distributed over gloo).
//
// Round 4 -- hardening before the code meets more than one GPU:
//   * the prototypes are RCCL's own (decltype of the header's declarations), the datatype is ncclFloat32 by name;
//   * the loaded library's ncclGetVersion must have the header's major version;
//   * after ncclCommInitRank the communicator is asked what IT thinks: ncclCommCount / ncclCommUserRank must agree with the
//     arguments, and bench.py reports that count (collective.ranks_seen_by_rccl), not what the launcher said;
//   * nothing waits for ever (KWS_COMM_TIMEOUT_MS, default 120 000): ncclCommInitRank -- which returns when every rank has joined --
//     runs on a helper thread that the caller waits for against the deadline (a rank that never shows up makes the others fail with
//     KWS_ERROR_HIP instead of hanging the job); the all-gather only enqueues, and kws_comm_wait polls the stream and
//     ncclCommGetAsyncError against the same deadline and aborts the communicator (ncclCommAbort) when it passes or a peer has failed.
//     The communicator itself stays a blocking one: the per-step enqueue is RCCL's ordinary path.
#include "kws_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

static_assert(NCCL_UNIQUE_ID_BYTES == KWS_COMM_ID_BYTES, "kws.h publishes the size of ncclUniqueId");

namespace {
struct Rccl {
    void *lib = nullptr;
    int version = 0;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int timeout_ms()
{
    const char *ev = getenv("KWS_COMM_TIMEOUT_MS");
    const int v = ev ? atoi(ev) : 0;
    return v > 0 ? v : 120000;
}

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
#define KWS_RCCL_SYM(field, name) r.field = (decltype(r.field))dlsym(lib, name)
    KWS_RCCL_SYM(GetVersion, "ncclGetVersion");
    KWS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    KWS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    KWS_RCCL_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
    KWS_RCCL_SYM(CommCount, "ncclCommCount");
    KWS_RCCL_SYM(CommUserRank, "ncclCommUserRank");
    KWS_RCCL_SYM(AllGather, "ncclAllGather");
    KWS_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    KWS_RCCL_SYM(CommAbort, "ncclCommAbort");
    KWS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef KWS_RCCL_SYM
    if (!r.GetVersion || !r.GetUniqueId || !r.CommInitRank || !r.CommGetAsyncError || !r.CommCount || !r.CommUserRank || !r.AllGather ||
        !r.CommDestroy || !r.CommAbort || !r.GetErrorString) {
        dlclose(lib);
        return fail(KWS_ERROR_HIP, "librccl lacks one of the entry points kws_comm.cpp binds (ncclGetVersion, ncclCommInitRank, ncclCommGetAsyncError, ncclCommCount, ...)");
    }
    // NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch since 2.9: the ABI this file was compiled against is the header's major version
    if (r.GetVersion(&r.version) != ncclSuccess || r.version / 10000 != NCCL_MAJOR) {
        const int v = r.version;
        dlclose(lib);
        return fail(KWS_ERROR_HIP, "librccl reports version %d; this library was built against the RCCL %d.x API (rccl.h %d)", v, NCCL_MAJOR, NCCL_VERSION_CODE);
    }
    r.lib = lib;
    g_rccl = r;
    return EI_IMPULSE_OK;
}
}  // namespace

struct kws_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    int seen_world = 0, seen_rank = -1;         // what RCCL itself reports for this communicator
    bool aborted = false;
};

namespace {
// Polls until everything enqueued on `stream` has completed, a peer has failed, or the deadline has passed (then: abort).
EI_IMPULSE_ERROR settle(kws_comm *c, const char *what, hipStream_t stream)
{
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms());
    for (;;) {
        ncclResult_t st = ncclSuccess;
        const ncclResult_t r = g_rccl.CommGetAsyncError(c->comm, &st);
        if (r != ncclSuccess) return fail(KWS_ERROR_HIP, "%s: ncclCommGetAsyncError: %s", what, g_rccl.GetErrorString(r));
        if (st != ncclSuccess && st != ncclInProgress) {
            (void)g_rccl.CommAbort(c->comm);
            c->aborted = true;
            return fail(KWS_ERROR_HIP, "%s (rank %d of %d): %s", what, c->rank, c->world, g_rccl.GetErrorString(st));
        }
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return EI_IMPULSE_OK;
        if (q != hipErrorNotReady) {
            // the stream itself failed: a later ncclCommDestroy would wait on it -- abort here too
            (void)g_rccl.CommAbort(c->comm);
            c->aborted = true;
            return fail(KWS_ERROR_HIP, "%s: %s; communicator aborted", what, hipGetErrorString(q));
        }
        if (std::chrono::steady_clock::now() > deadline) {
            (void)g_rccl.CommAbort(c->comm);
            c->aborted = true;
            return fail(KWS_ERROR_HIP, "%s (rank %d of %d): no progress within %d ms -- is a rank missing?  communicator aborted", what, c->rank, c->world, timeout_ms());
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

// ncclCommInitRank on a helper thread: it returns when every rank has joined, and there is no handle to abort before it has
struct InitJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    bool abandoned = false;        // the caller gave up at its deadline: a communicator that still comes into being belongs to nobody
    ncclResult_t result = ncclSuccess;
    ncclComm_t comm = nullptr;
};
}  // namespace

#pragma GCC visibility push(default)
extern "C" {

EI_IMPULSE_ERROR kws_comm_unique_id(void *id, size_t nbytes)
{
    if (!id || nbytes < KWS_COMM_ID_BYTES) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_comm_unique_id needs a %d-byte buffer", KWS_COMM_ID_BYTES);
    EI_IMPULSE_ERROR e = load_rccl();
    if (e) return e;
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return fail(KWS_ERROR_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
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
    ncclUniqueId u;
    memcpy(u.internal, id, KWS_COMM_ID_BYTES);
    kws_comm *c = new kws_comm();
    c->world = world_size; c->rank = rank; c->device = device;
    {
        auto job = std::make_shared<InitJob>();
        std::thread([job, world_size, u, rank, device]() {
            ncclComm_t comm = nullptr;
            ncclResult_t r = hipSetDevice(device) == hipSuccess ? g_rccl.CommInitRank(&comm, world_size, u, rank) : ncclUnhandledCudaError;
            std::lock_guard<std::mutex> lk(job->mu);
            job->result = r; job->comm = comm; job->done = true;
            // a peer that joins after the caller's deadline completes the initialisation into a communicator nobody holds: abort it here
            // instead of leaking it
            if (job->abandoned && r == ncclSuccess && comm) { (void)g_rccl.CommAbort(comm); job->comm = nullptr; }
            job->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(job->mu);
        if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms()), [&] { return job->done; })) {
            job->abandoned = true;      // (under job->mu) the helper stays blocked inside RCCL and holds its own reference to the job
            delete c;
            return fail(KWS_ERROR_HIP, "ncclCommInitRank(rank %d of %d): not every rank joined within %d ms", rank, world_size, timeout_ms());
        }
        if (job->result != ncclSuccess) { const ncclResult_t r = job->result; delete c; return fail(KWS_ERROR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world_size, g_rccl.GetErrorString(r)); }
        c->comm = job->comm;
    }
    // what the communicator itself says it is
    if (g_rccl.CommCount(c->comm, &c->seen_world) != ncclSuccess || g_rccl.CommUserRank(c->comm, &c->seen_rank) != ncclSuccess ||
        c->seen_world != world_size || c->seen_rank != rank) {
        const int sw = c->seen_world, sr = c->seen_rank;
        (void)g_rccl.CommAbort(c->comm);
        delete c;
        return fail(KWS_ERROR_HIP, "RCCL reports rank %d of %d for a communicator created as rank %d of %d", sr, sw, rank, world_size);
    }
    *out = c;
    return EI_IMPULSE_OK;
}

int kws_comm_world_size(const kws_comm *c) { return c ? c->world : 0; }
int kws_comm_rank(const kws_comm *c) { return c ? c->rank : -1; }
int kws_comm_ranks_seen(const kws_comm *c) { return c ? c->seen_world : 0; }
int kws_comm_rccl_version(void) { return load_rccl() == EI_IMPULSE_OK ? g_rccl.version : 0; }

EI_IMPULSE_ERROR kws_allgather_scores(kws_comm *c, const float *local_scores, float *all_scores, size_t clips_per_rank, int label_count,
                                      void *stream)
{
    if (!c || !local_scores || !all_scores || label_count < 1) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_allgather_scores: bad argument");
    if (c->aborted) return fail(KWS_ERROR_HIP, "kws_allgather_scores: the communicator was aborted");
    if (clips_per_rank == 0) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(c->device));
    const ncclResult_t r = g_rccl.AllGather(local_scores, all_scores, clips_per_rank * (size_t)label_count, ncclFloat32, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(KWS_ERROR_HIP, "ncclAllGather: %s", g_rccl.GetErrorString(r));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_comm_wait(kws_comm *c, void *stream)
{
    if (!c) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_comm_wait: null communicator");
    if (c->aborted) return fail(KWS_ERROR_HIP, "kws_comm_wait: the communicator was aborted");
    HIP_TRY(hipSetDevice(c->device));
    return settle(c, "waiting for the collectives on the stream", (hipStream_t)stream);
}

void kws_comm_destroy(kws_comm *c)
{
    if (!c) return;
    if (c->comm && !c->aborted && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
#pragma GCC visibility pop
