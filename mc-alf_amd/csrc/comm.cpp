// libmcalf_hip.so, host side: the in-library gather of the per-sample logL shards over RCCL (one process per GPU).
#include <dlfcn.h>

#include <mutex>

#include "host_ctx.h"

// ---------------------------------------------------------------------------------------------------------
// Multi-GPU inside the library: one process per GPU, each with its own context; the context
// owns an RCCL communicator and the per-sample logL shards travel to the root rank as ONE grouped send/receive
// exchange (what ncclGather is) enqueued on the launch stream right behind the kernels -- no host round trip, no
// Python in the step.  RCCL is resolved with dlopen at the first mcalf_comm_* call (torch's bundled librccl when
// torch is in the process, the ROCm one otherwise), so the single-GPU path has no dependency on it.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi g_rccl;

std::mutex g_rccl_mutex;

// (serialised: contexts of several host threads may reach their first mcalf_comm_* call together; a failed load is
// retried by the next call)
int rccl_load(mcalf_ctx* ctx) {
    std::lock_guard<std::mutex> guard(g_rccl_mutex);
    if (g_rccl.ok) return MCALF_OK;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    // MCALF_RCCL_LIB: an explicit library (tests put a two-process stand-in here to drive the N > 1 branch on a
    // one-GPU box; see tests/stubs/)
    // (the context's snapshot of the environment; mcalf_comm_unique_id has no context and takes one of its own)
    const std::string over = ctx ? ctx->env.rccl_lib : read_environment().rccl_lib;
    if (!over.empty() && !(h = dlopen(over.c_str(), RTLD_NOW | RTLD_LOCAL)))
        return set_err(ctx, MCALF_ERR_COMM, "MCALF_RCCL_LIB=%s: %s", over.c_str(), dlerror());
    for (int i = 0; !h && i < 2; ++i)                    // a copy already in the process (torch's) wins
        h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (int i = 0; !h && i < 2; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_err(ctx, MCALF_ERR_COMM, "librccl.so not found: %s", dlerror());
    g_rccl.handle = h;
#define MCALF_SYM(field, sym)                                                                     \
    if (!(g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym))))                \
        return set_err(ctx, MCALF_ERR_COMM, "librccl.so lacks %s", sym);
    MCALF_SYM(GetUniqueId, "ncclGetUniqueId")
    MCALF_SYM(CommInitRank, "ncclCommInitRank")
    MCALF_SYM(CommDestroy, "ncclCommDestroy")
    MCALF_SYM(CommAbort, "ncclCommAbort")
    MCALF_SYM(Send, "ncclSend")
    MCALF_SYM(Recv, "ncclRecv")
    MCALF_SYM(GroupStart, "ncclGroupStart")
    MCALF_SYM(GroupEnd, "ncclGroupEnd")
    MCALF_SYM(GetErrorString, "ncclGetErrorString")
#undef MCALF_SYM
    g_rccl.ok = true;
    return MCALF_OK;
}
}  // namespace

#define RCCL_TRY(ctx, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return set_err(ctx, MCALF_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));           \
    } while (0)

void comm_release(mcalf_ctx* ctx) {
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    if (ctx->comm && g_rccl.ok) (void)(ctx->comm_dead ? g_rccl.CommAbort(ctx->comm) : g_rccl.CommDestroy(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
    ctx->comm_rank = -1;
    ctx->comm_dead = false;
    ctx->comm_calls = 0;
    ctx->ev_comm_used[0] = ctx->ev_comm_used[1] = false;
}

// A failure inside an exchange leaves the ranks out of step: the communicator is aborted (outstanding RCCL work
// is torn down instead of waiting for peers that will never call) and every later gather on it is refused.
static int comm_fail(mcalf_ctx* ctx, const char* what, ncclResult_t r) {
    const int rc = set_err(ctx, MCALF_ERR_COMM, "%s failed: %s; the communicator has been aborted -- call mcalf_comm_destroy / "
                           "mcalf_comm_init on every rank before the next gather", what, g_rccl.GetErrorString(r));
    if (ctx->comm && !ctx->comm_dead) {
        (void)g_rccl.CommAbort(ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_dead = true;
    return rc;
}

extern "C" int mcalf_comm_unique_id(void* id128) {
    if (!id128) return set_err(nullptr, MCALF_ERR_INVALID, "id is NULL");
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == MCALF_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    RCCL_TRY(nullptr, g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return MCALF_OK;
}

extern "C" int mcalf_comm_init(mcalf_ctx* ctx, const void* id128, int32_t nranks, int32_t rank) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
        return set_err(ctx, MCALF_ERR_INVALID, "mcalf_comm_init: bad arguments (nranks %d, rank %d)", nranks, rank);
    MCALF_SINGLE_ONLY(ctx, "mcalf_comm_init (the communicator joins one process per GPU; a multi-device context needs none)");
    int rc = rccl_load(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    RCCL_TRY(ctx, g_rccl.CommInitRank(&ctx->comm, nranks, id, rank));      // collective over all ranks
    ctx->comm_ranks = nranks;
    ctx->comm_rank = rank;
    return MCALF_OK;
}

extern "C" int mcalf_comm_info(const mcalf_ctx* ctx, int32_t* nranks, int32_t* rank) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (nranks) *nranks = ctx->comm_ranks;
    if (rank) *rank = ctx->comm_rank;
    return MCALF_OK;
}

extern "C" int mcalf_comm_destroy(mcalf_ctx* ctx) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    MCALF_SINGLE_ONLY(ctx, "mcalf_comm_destroy");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    return MCALF_OK;
}

extern "C" int mcalf_comm_set_overlap(mcalf_ctx* ctx, int32_t on) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    ctx->comm_overlap = on ? 1 : 0;
    return MCALF_OK;
}

extern "C" int mcalf_comm_join(mcalf_ctx* ctx, void* stream) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    MCALF_SINGLE_ONLY(ctx, "mcalf_comm_join");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < 2; ++k)
        if (ctx->ev_comm_used[k]) HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_comm[k], 0));
    return MCALF_OK;
}

static int comm_ensure_streams(mcalf_ctx* ctx) {
    if (!ctx->comm_stream) { const int rcs = create_stream(ctx, &ctx->comm_stream); if (rcs) return rcs; }   // (with the context's CU mask, when it has one)
    if (!ctx->ev_kernels) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_kernels, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->ev_comm)
        if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return MCALF_OK;
}

extern "C" int mcalf_loglike_gatherv_device(mcalf_ctx* ctx, const double* dP, int64_t batch_local, double* dlogL_local,
                                            double* dlogL_all, const int64_t* counts, int32_t root, void* stream) {
    // ---- 1. argument checks: a non-zero return from here means NOTHING was enqueued on this rank ----------
    if (!ctx || batch_local < 0 || (batch_local > 0 && (!dP || !dlogL_local)))
        return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    if (ctx->comm_dead) return set_err(ctx, MCALF_ERR_COMM, "the communicator was aborted after a failed exchange; re-initialise it");
    if (!ctx->comm) return set_err(ctx, MCALF_ERR_INVALID, "mcalf_comm_init has not been called");
    const int nranks = ctx->comm_ranks, me = ctx->comm_rank;
    if (root < 0 || root >= nranks) return set_err(ctx, MCALF_ERR_INVALID, "root %d out of range", root);
    if (counts && counts[me] != batch_local)
        return set_err(ctx, MCALF_ERR_INVALID, "counts[%d] = %lld but batch_local = %lld", me, (long long)counts[me],
                       (long long)batch_local);
    const bool is_root = me == root;
    int64_t total = 0, my_off = 0;
    for (int r = 0; r < nranks; ++r) {
        const int64_t c = counts ? counts[r] : batch_local;
        if (c < 0) return set_err(ctx, MCALF_ERR_INVALID, "counts[%d] is negative", r);
        if (r < me) my_off += c;
        total += c;
    }
    if (is_root && total > 0 && !dlogL_all) return set_err(ctx, MCALF_ERR_INVALID, "root needs dlogL_all");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)batch_local;
    ctx->last.path = MCALF_PATH_DEVICE; ctx->last.pinned_in = ctx->last.pinned_out = 0;

    // ---- 2. everything that can fail locally, BEFORE anything is enqueued ---------------------------------
    // A rank that fails here (or in the kernel launches below) still takes its part in the exchange, with a block
    // of NaNs, and reports its error afterwards: the peers' sends / receives complete and the root sees which rows
    // are missing, instead of waiting in ncclRecv for a send that never comes.
    int rc_local = launch_preflight(ctx, kModeLogL, batch_local);
    // The exchange runs on the launch stream itself by default (nothing crosses streams: the call has plain stream
    // semantics for free) and, in overlap mode with more than one rank, on the context's exchange stream behind an
    // event of the launch stream, so that the root's NEXT kernels do not queue behind receives that wait for its
    // peers.  (The event traffic of that mode costs about 20 us of queue time per step, measured on one GPU.)
    const bool side = nranks > 1 && ctx->comm_overlap;
    if (side) {
        const int rs = comm_ensure_streams(ctx);
        if (rs != MCALF_OK) return comm_fail(ctx, "creating the exchange stream / events", ncclSystemError);
    }
    const unsigned slot = ctx->comm_calls & 1u;
    // the exchange that used this slot two calls ago read the caller's buffers of that call: it must have
    // landed before this call's kernels overwrite them (a caller in overlap mode alternates two buffer pairs)
    if (side && ctx->ev_comm_used[slot]) {
        if (hipStreamWaitEvent(st, ctx->ev_comm[slot], 0) != hipSuccess)
            return comm_fail(ctx, "hipStreamWaitEvent", ncclSystemError);
    }
    // ---- 3. kernels ------------------------------------------------------------------------------------------
    if (rc_local == MCALF_OK && n > 0) rc_local = launch(ctx, kModeLogL, dP, batch_local, 0, 0, dlogL_local, nullptr, st);
    const std::string local_msg = ctx->err;
    if (rc_local != MCALF_OK && n > 0) (void)hipMemsetAsync(dlogL_local, 0xFF, n * sizeof(double), st);   // NaN block
    // ---- 4. exchange -----------------------------------------------------------------------------------------
    hipStream_t cs = side ? ctx->comm_stream : st;
    if (side && (hipEventRecord(ctx->ev_kernels, st) != hipSuccess || hipStreamWaitEvent(cs, ctx->ev_kernels, 0) != hipSuccess))
        return comm_fail(ctx, "hipEventRecord / hipStreamWaitEvent", ncclSystemError);
    if (is_root) {
        // (one rank: the "gather" is this device-to-device copy behind the kernels)
        if (n > 0 && hipMemcpyAsync(dlogL_all + my_off, dlogL_local, n * sizeof(double), hipMemcpyDeviceToDevice, cs) != hipSuccess)
            return comm_fail(ctx, "hipMemcpyAsync", ncclSystemError);
        if (nranks > 1) {
            ncclResult_t e = g_rccl.GroupStart();
            int64_t off = 0;
            for (int r = 0; r < nranks && e == ncclSuccess; ++r) {
                const int64_t c = counts ? counts[r] : batch_local;
                if (r != root && c > 0) e = g_rccl.Recv(dlogL_all + off, (size_t)c, ncclFloat64, r, ctx->comm, cs);
                off += c;
            }
            const ncclResult_t e2 = g_rccl.GroupEnd();
            if (e != ncclSuccess || e2 != ncclSuccess) return comm_fail(ctx, "ncclRecv group", e != ncclSuccess ? e : e2);
        }
    } else if (n > 0) {
        const ncclResult_t e = g_rccl.Send(dlogL_local, n, ncclFloat64, root, ctx->comm, cs);
        if (e != ncclSuccess) return comm_fail(ctx, "ncclSend", e);
    }
    if (side) {
        if (hipEventRecord(ctx->ev_comm[slot], cs) != hipSuccess) return comm_fail(ctx, "hipEventRecord", ncclSystemError);
        ctx->ev_comm_used[slot] = true;
    }
    ctx->comm_calls++;
    if (rc_local != MCALF_OK) {
        ctx->err = local_msg + " (this rank sent a block of NaNs so that the exchange completes)";
        set_last_error(ctx->err);
    }
    return rc_local;
}

extern "C" int mcalf_loglike_gather_device(mcalf_ctx* ctx, const double* dP, int64_t batch_local, double* dlogL_local,
                                           double* dlogL_all, int32_t root, void* stream) {
    return mcalf_loglike_gatherv_device(ctx, dP, batch_local, dlogL_local, dlogL_all, nullptr, root, stream);
}

