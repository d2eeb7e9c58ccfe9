// libmcalf_hip.so, host side: the one-theta-at-a-time serving layer -- the resident evaluator of a context
// (mcalf_set_resident) and the likelihood broker's serving loops (mcalf_broker_serve, mcalf_broker_serve_resident).
// Opt-in and frozen: nothing here is on the path of a batch call.
#include "host_ctx.h"

// ---- resident one-theta evaluator, host side (device side: mcalf_resident_kernel) -------------------------------------
// Tell the resident kernel to leave and wait until it has (bounded by its own idle limit).
void resident_stop(mcalf_ctx* ctx) {
    if (!ctx->h_box || !ctx->res_stream) return;
    if (ctx->res_alive) {
        __atomic_store_n(&ctx->h_box->quit, 1u, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(ctx->res_stream);
        ctx->res_alive = false;
    }
}

bool resident_serves(const mcalf_ctx* ctx, int mode, int64_t batch, int rowlen, bool from_cube) {
    // (a wide-LSF context convolves in a second kernel: its calls launch)
    return ctx->resident_us > 0 && mode == kModeLogL && batch == 1 && ctx->ntiles == 1 && !ctx->wide && rowlen <= kResRowMax && !from_cube &&
           !ctx->profiling;
}

// One theta through the resident kernel: post the request (launching the kernel when there is none), spin on the result.
int resident_call(mcalf_ctx* ctx, const double* row, int rowlen, double* out) {
    if (!ctx->h_box) {
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_box, sizeof(ResidentBox), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset((void*)ctx->h_box, 0, sizeof(ResidentBox));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->d_box, (void*)ctx->h_box, 0));
        { const int rcs = create_stream(ctx, &ctx->res_stream); if (rcs) return rcs; }      // (with the context's CU mask, when it has one)
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_res_shared, sizeof(ResidentShared)));
        const void* k = resident_kernel_ptr(ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX, ctx->selfhalo != 0);
        HIP_TRY(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsBudget + 1024)));
    }
    ResidentBox* box = ctx->h_box;
    uint64_t* res = reinterpret_cast<uint64_t*>(&box->result);
    __atomic_store_n(res, kResultPending, __ATOMIC_RELAXED);
    std::memcpy(box->row, row, (size_t)rowlen * sizeof(double));
    const unsigned seq = ++ctx->res_seq;
    ctx->last.path = MCALF_PATH_HOST_ZEROCOPY; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    ctx->last.persistent = 0; ctx->last.grid = 1; ctx->last.items = 1; ctx->last.inline_setup = 3; ctx->last.stream_polled = 1;
    ctx->res_calls++;
    auto launch_kernel = [&]() -> int {
        // (a previous kernel of this context has said "gone", or there was none: a new one starts behind it on the stream)
        KArgs a = make_kargs(ctx, kModeLogL, nullptr, 0, 1, 0, 0, 0, nullptr, nullptr, false, nullptr);
        a.persist = 0; a.order = nullptr;
        __atomic_store_n(&box->state, kResRunning, __ATOMIC_RELAXED);
        __atomic_store_n(&box->quit, 0u, __ATOMIC_RELAXED);
        __atomic_store_n(&box->req, seq, __ATOMIC_RELEASE);
        ResidentBox* dbox = ctx->d_box;
        ResidentShared* dsh = ctx->d_res_shared;
        long long idle = (long long)ctx->resident_us * 100;                  // ticks of the 100 MHz clock
        int row_off = (int)((ctx->lds_bytes_inline / sizeof(double) + 1) & ~(size_t)1);
        void* kargs[] = {(void*)&a, (void*)&dbox, (void*)&dsh, (void*)&idle, (void*)&row_off};
        const size_t lds = (size_t)row_off * sizeof(double) + kResRowMax * sizeof(double) + 16;
        HIP_TRY(ctx, hipMemsetAsync(dsh, 0, sizeof(ResidentShared), ctx->res_stream));
        HIP_TRY(ctx, hipLaunchKernel(resident_kernel_ptr(ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX, ctx->selfhalo != 0), dim3(1), dim3(kBlock),
                                     kargs, lds, ctx->res_stream));
        ctx->res_alive = true;
        ctx->res_launches++;
        return MCALF_OK;
    };
    int rc;
    if (!ctx->res_alive) { if ((rc = launch_kernel())) return rc; }
    else __atomic_store_n(&box->req, seq, __ATOMIC_RELEASE);
    const double t0 = now_us();
    for (unsigned long spins = 1;; ++spins) {
        // answered: the result slot no longer holds the pending pattern -- or, should the answer ITSELF be that bit pattern (a
        // theta carrying the NaN payload through the arithmetic), the acknowledgement says so: the kernel stores the result,
        // waits for the store, then writes `ack` (same cache line: the second look costs nothing)
        if (__atomic_load_n(res, __ATOMIC_ACQUIRE) != kResultPending || __atomic_load_n(&box->ack, __ATOMIC_ACQUIRE) == seq) break;
        if (__atomic_load_n(&box->state, __ATOMIC_ACQUIRE) == kResGone) {
            // the kernel left without having seen this request (it looks once more after saying "leaving", so a request it
            // has seen is answered): a new one takes it
            if (__atomic_load_n(res, __ATOMIC_ACQUIRE) != kResultPending || __atomic_load_n(&box->ack, __ATOMIC_ACQUIRE) == seq) break;
            ctx->res_alive = false;
            if ((rc = launch_kernel())) return rc;
        }
        if ((spins & 0xFFFFul) == 0) {
            const hipError_t q = hipStreamQuery(ctx->res_stream);
            if (q != hipSuccess && q != hipErrorNotReady) {
                ctx->res_alive = false;
                return set_err(ctx, MCALF_ERR_HIP, "resident evaluator failed: %s", hipGetErrorString(q));
            }
            if (now_us() - t0 > 5e6) {
                ctx->res_alive = false;
                return set_err(ctx, MCALF_ERR_HIP, "resident evaluator did not answer within 5 s");
            }
        }
        __builtin_ia32_pause();
    }
    *out = box->result;
    return MCALF_OK;
}

extern "C" int mcalf_set_resident(mcalf_ctx* ctx, int32_t idle_us) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (idle_us < 0 || idle_us > 1000000) return set_err(ctx, MCALF_ERR_INVALID, "idle limit must be 0 (off) .. 1000000 us");
    if (is_multi(ctx)) {                                  // one-theta calls of a multi-device context go to its first device
        const int rc = mcalf_set_resident(ctx->subs[0], idle_us);
        return rc ? set_err(ctx, rc, "%s", ctx->subs[0]->err.c_str()) : MCALF_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (idle_us == 0) resident_stop(ctx);
    ctx->resident_us = idle_us;
    return MCALF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Likelihood broker, serving side (many one-theta-at-a-time solver ranks on one GPU; mc-alf_amd/broker.py holds the
// shared-memory protocol and the ranks' side).  The reference's solvers call the likelihood one theta at a time, one
// MPI rank per core (cli.py:37-41, 110; hires_fitter.py:250-262); launches of DIFFERENT processes do not overlap
// beyond a few, launches of ONE process on several streams do.  So one thread of one process serves every rank: it
// collects the open requests, evaluates them as one small batch on a context that is free (the one-launch variant of
// small calls, parameters and results in page-locked memory, nothing synchronous) and goes on polling -- requests that
// arrive while a launch is in flight leave at once on the next free context instead of waiting for it to end.
// A live point's value does not depend on the batch it is evaluated in, so every rank gets the bits its own context
// would give.
// ---------------------------------------------------------------------------------------------------------
extern "C" int mcalf_broker_serve(mcalf_ctx* const* ctxs, int32_t nctx, const mcalf_broker_t* b, double max_seconds) {
    constexpr int kMaxLanes = 8;
    if (!ctxs || nctx < 1 || nctx > kMaxLanes || !ctxs[0]) return set_err(nullptr, MCALF_ERR_INVALID, "broker: 1 .. %d contexts", kMaxLanes);
    mcalf_ctx* c0 = ctxs[0];
    if (!b || b->slots < 1 || b->slots > 65536 || !b->req || !b->ack || !b->theta || !b->logl || !b->stop || b->counter_stride < 1 ||
        b->theta_stride < b->ndim || b->logl_stride < 1)
        return set_err(c0, MCALF_ERR_INVALID, "broker: incomplete description of the request block");
    struct Lane { mcalf_ctx* c; int n; bool busy; unsigned long polls; std::vector<int> slot; std::vector<uint64_t> seq; };
    // Completion of a launch is read off its results: the slots are filled with a NaN no kernel produces before the
    // launch, and the launch is over when none is left (the results land in page-locked memory; asking the runtime --
    // hipStreamQuery in a loop -- cost more per round than the Python loop's blocking wait).  The stream is asked only
    // now and then, so that a failed launch cannot keep the loop waiting.
    constexpr uint64_t kPending = kResultPending;
    Lane lane[kMaxLanes];
    int rc;
    for (int k = 0; k < nctx; ++k) {
        mcalf_ctx* c = ctxs[k];
        if (!c || c->ndim != b->ndim) return set_err(c0, MCALF_ERR_INVALID, "broker: context %d does not take rows of %d parameters", k, b->ndim);
        MCALF_SINGLE_ONLY(c, "mcalf_broker_serve (list one single-device context per lane)");
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == c) return set_err(c0, MCALF_ERR_INVALID, "broker: context %d is listed twice (a context holds one batch at a time)", k);
        HIP_TRY(c, hipSetDevice(c->device));
        if ((rc = ensure_small(c))) return rc;
        lane[k].c = c; lane[k].n = 0; lane[k].busy = false; lane[k].polls = 0;
    }
    const int ndim = b->ndim, slots = b->slots;
    const int cap = (int)std::min<size_t>((size_t)slots, kSmallDoubles / (size_t)ndim);    // live points per launch
    std::vector<unsigned char> inflight((size_t)slots, 0);
    const double t_begin = now_us();
    double t_last = t_begin;
    bool stopping = false;
    auto finish = [&](Lane& L) {                          // results first, then the acknowledgement the rank is polling
        for (int i = 0; i < L.n; ++i) b->logl[(size_t)L.slot[i] * b->logl_stride] = L.c->h_small[kSmallDoubles + i];
        for (int i = 0; i < L.n; ++i) {
            __atomic_store_n(const_cast<uint64_t*>(b->ack + (size_t)L.slot[i] * b->counter_stride), L.seq[i], __ATOMIC_RELEASE);
            inflight[(size_t)L.slot[i]] = 0;
        }
        if (b->stats) { b->stats[0] += 1; b->stats[1] += (uint64_t)L.n; }
        L.busy = false;
    };
    while (true) {
        bool progress = false, any_busy = false;
        int free_lane = -1;
        for (int k = 0; k < nctx; ++k) {
            Lane& L = lane[k];
            if (L.busy) {
                const uint64_t* res = reinterpret_cast<const uint64_t*>(L.c->h_small + kSmallDoubles);
                bool done = true;
                for (int i = L.n - 1; i >= 0 && done; --i) done = __atomic_load_n(res + i, __ATOMIC_ACQUIRE) != kPending;
                if (done) { finish(L); progress = true; }
                else if ((++L.polls & 0x3FFFul) == 0) {
                    const hipError_t q = hipStreamQuery(L.c->stream);
                    if (q == hipSuccess) {                 // (the stream has drained: every result must be there now)
                        for (int i = 0; i < L.n; ++i)
                            if (__atomic_load_n(res + i, __ATOMIC_ACQUIRE) == kPending)
                                return set_err(L.c, MCALF_ERR_HIP, "broker: a launch ended without its results");
                    } else if (q != hipErrorNotReady) {
                        return set_err(L.c, MCALF_ERR_HIP, "broker: launch failed: %s", hipGetErrorString(q));
                    }
                }
            }
            if (L.busy) any_busy = true;
            else if (free_lane < 0) free_lane = k;
        }
        if (__atomic_load_n(const_cast<uint64_t*>(b->stop), __ATOMIC_ACQUIRE) != 0) stopping = true;
        if (stopping) {
            if (!any_busy) return MCALF_OK;               // (what was in flight has been answered)
            continue;
        }
        if (free_lane >= 0) {
            Lane& L = lane[free_lane];
            L.slot.clear(); L.seq.clear();
            for (int s = 0; s < slots && (int)L.slot.size() < cap; ++s) {
                if (inflight[(size_t)s]) continue;
                const uint64_t r = __atomic_load_n(const_cast<uint64_t*>(b->req + (size_t)s * b->counter_stride), __ATOMIC_ACQUIRE);
                if (r != b->ack[(size_t)s * b->counter_stride]) { L.slot.push_back(s); L.seq.push_back(r); }
            }
            L.n = (int)L.slot.size();
            if (L.n > 0) {
                for (int i = 0; i < L.n; ++i)
                    std::memcpy(L.c->h_small + (size_t)i * ndim, b->theta + (size_t)L.slot[i] * b->theta_stride, (size_t)ndim * sizeof(double));
                uint64_t* res = reinterpret_cast<uint64_t*>(L.c->h_small + kSmallDoubles);
                for (int i = 0; i < L.n; ++i) __atomic_store_n(res + i, kPending, __ATOMIC_RELEASE);
                L.polls = 0;
                L.c->last.path = MCALF_PATH_HOST_ZEROCOPY; L.c->last.pinned_in = L.c->last.pinned_out = 0;
                if ((rc = launch(L.c, kModeLogL, L.c->d_small, L.n, 0, 0, L.c->d_small + kSmallDoubles, nullptr, L.c->stream))) {
                    for (int k = 0; k < nctx; ++k)
                        if (lane[k].busy) (void)hipStreamSynchronize(lane[k].c->stream);
                    return rc;
                }
                for (int i = 0; i < L.n; ++i) inflight[(size_t)L.slot[i]] = 1;
                L.busy = true;
                progress = true;
            }
        }
        const double t = now_us();
        if (progress) t_last = t;
        else if (!any_busy && (t - t_last) * 1e-6 > b->idle_sleep_after_s) {
            struct timespec ts = {0, 200000};             // nobody has asked for a while: yield the core between polls
            nanosleep(&ts, nullptr);
        }
        if (max_seconds > 0 && (t - t_begin) * 1e-6 > max_seconds) stopping = true;
    }
}

// The broker with resident evaluators: one workgroup per solver rank, polling the rank's mailbox in the shared block.
// The serving thread is OFF the path of a call: it only starts the launch of those workgroups when a request finds none.
// CO-RESIDENCY: a mailbox is served only while ITS workgroup is on the chip, and the launch's workgroups are all expected
// to be there at once -- one 145-register workgroup per compute unit, hence `slots` <= the device's CU count (taken from
// hipDeviceProp_t at mcalf_create).  Another kernel that occupies CUs (this process's batch calls, another process) can
// keep some of the launch's workgroups waiting for a CU: their mailboxes are not polled until one frees up, and a rank's
// call waits that long -- it is never lost (the request stays in the mailbox).
extern "C" int mcalf_broker_serve_resident(mcalf_ctx* ctx, void* boxes, int32_t slots, volatile uint64_t* stop, int32_t idle_us,
                                           uint64_t* stats, double max_seconds) {
    static_assert(sizeof(ResidentBox) == MCALF_MAILBOX_BYTES, "mailbox layout of include/mcalf_hip.h");
    static_assert(kResultPending == MCALF_RESULT_PENDING, "pending pattern of include/mcalf_hip.h");
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    MCALF_SINGLE_ONLY(ctx, "mcalf_broker_serve_resident");
    if (!boxes || !stop || slots < 1 || slots > ctx->num_cu || idle_us < 1 || idle_us > 1000000 || (reinterpret_cast<uintptr_t>(boxes) & 63))
        return set_err(ctx, MCALF_ERR_INVALID, "resident broker: 1 .. %d mailboxes (one co-resident workgroup per compute unit of this "
                       "device) at a 64-byte aligned address, idle limit 1 .. 1000000 us", ctx->num_cu);
    if (ctx->ntiles != 1 || ctx->wide || ctx->ndim > kResRowMax)
        return set_err(ctx, MCALF_ERR_RANGE, "resident broker: the spectrum must fit one pixel tile WITH its LSF halo and a row 64 "
                       "parameters (%d tiles, %d parameters%s): use mcalf_broker_serve", ctx->ntiles, ctx->ndim,
                       ctx->wide ? ", LSF wider than a tile" : "");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ResidentBox* hb = static_cast<ResidentBox*>(boxes);
    const size_t bytes = (size_t)slots * sizeof(ResidentBox);
    HIP_TRY(ctx, hipHostRegister(boxes, bytes, hipHostRegisterMapped));
    ResidentBox* db = nullptr;
    ResidentShared* dsh = nullptr;
    hipStream_t st = nullptr;
    int rc = MCALF_OK;
    auto fail = [&](hipError_t e, const char* what) { rc = set_err(ctx, MCALF_ERR_HIP, "resident broker: %s failed: %s", what, hipGetErrorString(e)); };
    hipError_t he = hipHostGetDevicePointer((void**)&db, boxes, 0);
    if (he != hipSuccess) fail(he, "hipHostGetDevicePointer");
    const void* kern = resident_kernel_ptr(ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX, ctx->selfhalo != 0);
    if (rc == MCALF_OK && (he = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsBudget + 1024))) != hipSuccess)
        fail(he, "hipFuncSetAttribute");
    if (rc == MCALF_OK && (he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) fail(he, "hipStreamCreate");
    if (rc == MCALF_OK && (he = hipMalloc((void**)&dsh, sizeof(ResidentShared))) != hipSuccess) fail(he, "hipMalloc");
    bool launched = false;
    const double t_begin = now_us();
    while (rc == MCALF_OK) {
        if (__atomic_load_n(const_cast<uint64_t*>(stop), __ATOMIC_ACQUIRE) != 0) break;
        bool open = false;
        for (int s = 0; s < slots && !open; ++s)
            open = __atomic_load_n(&hb[s].req, __ATOMIC_ACQUIRE) != __atomic_load_n(&hb[s].ack, __ATOMIC_ACQUIRE);
        if (open) {
            // ONE launch serves every mailbox (workgroup k polls mailbox k) and its workgroups leave together: while it is
            // there, an open request is being answered -- or its workgroup has just left with the others and the launch is
            // about to end.  Only when the launch has ended is another one started.
            const hipError_t q = launched ? hipStreamQuery(st) : hipSuccess;
            if (q == hipSuccess) {
                KArgs a = make_kargs(ctx, kModeLogL, nullptr, 0, 1, 0, 0, 0, nullptr, nullptr, false, nullptr);
                a.persist = 0; a.order = nullptr;
                for (int s = 0; s < slots; ++s) __atomic_store_n(&hb[s].state, kResRunning, __ATOMIC_RELEASE);
                long long idle = (long long)idle_us * 100;
                int row_off = (int)((ctx->lds_bytes_inline / sizeof(double) + 1) & ~(size_t)1);
                void* kargs[] = {(void*)&a, (void*)&db, (void*)&dsh, (void*)&idle, (void*)&row_off};
                const size_t lds = (size_t)row_off * sizeof(double) + kResRowMax * sizeof(double) + 16;
                if ((he = hipMemsetAsync(dsh, 0, sizeof(ResidentShared), st)) != hipSuccess) { fail(he, "hipMemsetAsync"); break; }
                if ((he = hipLaunchKernel(kern, dim3((unsigned)slots), dim3(kBlock), kargs, lds, st)) != hipSuccess) { fail(he, "hipLaunchKernel"); break; }
                launched = true;
                if (stats) stats[0] += 1;
            } else if (q != hipErrorNotReady) {
                fail(q, "the resident launch");
                break;
            }
        }
        struct timespec ts = {0, 20000};                  // (the serving thread is not on a call's path: it only restarts the launch)
        nanosleep(&ts, nullptr);
        if (max_seconds > 0 && (now_us() - t_begin) * 1e-6 > max_seconds) break;
    }
    // everybody out: tell the workgroups, wait for them (bounded by the idle limit anyway), give the block back
    for (int s = 0; s < slots; ++s) __atomic_store_n(&hb[s].quit, 1u, __ATOMIC_RELEASE);
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int s = 0; s < slots; ++s) __atomic_store_n(&hb[s].quit, 0u, __ATOMIC_RELEASE);
    if (dsh) (void)hipFree(dsh);
    (void)hipHostUnregister(boxes);
    return rc;
}

