// libmcalf_hip.so, host side: the streaming single launch of the host-pointer entries (kernels.hip:
// mcalf_fused_kernel<..., kStream = true>).
#include "host_ctx.h"

// MCALF_STREAM_TRACE=1 (diagnostic): mean host-side microseconds per phase of the streaming entry (ctx->strace), printed when the
// context is destroyed

void stream_trace_report(mcalf_ctx* ctx) {
    if (!ctx->stream_trace || ctx->strace.n == 0) return;
    const double n = (double)ctx->strace.n;
    std::fprintf(stderr, "mcalf stream trace (%ld calls, us per call; workgroups per XCD in the last launch: %u .. %u): pointer checks %.2f, "
                 "prepare %.2f, launch %.2f, stage rows %.2f, wait %.2f, copy out %.2f\n", ctx->strace.n,
                 ctx->h_ctl ? ctx->h_ctl[2] : 0u, ctx->h_ctl ? ctx->h_ctl[3] : 0u, ctx->strace.t[0] / n, ctx->strace.t[1] / n, ctx->strace.t[2] / n,
                 ctx->strace.t[3] / n, ctx->strace.t[4] / n, ctx->strace.t[5] / n);
    ctx->strace = StreamTrace();
}

// Workspaces of the streaming launch for `batch` live points, and the words it shares with the host.
int stream_prepare(mcalf_ctx* ctx, int mode, int64_t batch) {
    int rc;
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    if (reduces && ctx->ntiles > 1 && (rc = grow(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)batch * ctx->ntiles * 4)))
        return rc;
    if ((size_t)batch > ctx->cap_sws) {
        // the launch's waves hand these to each other while it runs, through system-scope accesses (see the streaming
        // helpers); ordinary device memory
        // (every live point's records, taps and header start on a 128-byte line of their own)
        ctx->s_rec_stride = ((size_t)ctx->ncl_cap * kRecStride + 15) & ~(size_t)15;
        ctx->s_tap_stride = (2 * (size_t)ctx->n_cap + 8 + 15) & ~(size_t)15;
        ctx->s_hdr_stride = 16;
        const size_t nrec = (size_t)batch * ctx->s_rec_stride, ntap = (size_t)batch * ctx->s_tap_stride, nhdr = (size_t)batch * ctx->s_hdr_stride;
        const size_t npar = ((size_t)batch * ctx->ndim + 15) & ~(size_t)15;
        const size_t bytes = (nrec + ntap + nhdr + npar) * sizeof(double) + (size_t)batch * sizeof(unsigned int) + 256;
        if (ctx->d_sws) HIP_TRY(ctx, hipFree(ctx->d_sws));
        ctx->d_sws = nullptr; ctx->cap_sws = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_sws, bytes));
        HIP_TRY(ctx, hipMemset(ctx->d_sws, 0, bytes));                                      // (stamps start at 1, queues at 0)
        HIP_TRY(ctx, hipDeviceSynchronize());            // (the fill may still be running, and the launch streams do not wait for the null stream)
        ctx->s_recs = static_cast<double*>(ctx->d_sws);
        ctx->s_taps = ctx->s_recs + nrec;
        ctx->s_hdr = reinterpret_cast<SampleHdr*>(ctx->s_taps + ntap);
        ctx->s_P = ctx->s_taps + ntap + nhdr;
        ctx->d_ready = reinterpret_cast<unsigned int*>(ctx->s_P + npar);
        ctx->d_sctl = reinterpret_cast<StreamCtl*>((reinterpret_cast<uintptr_t>(ctx->d_ready + batch) + 63) & ~(uintptr_t)63);
        ctx->cap_sws = (size_t)batch;
        // (the generation count goes on: the new stamps are zero and no launch is ever stamped 0, while a count that
        // restarted could equal the completion word an earlier launch left in h_ctl[1])
    }
    return ensure_ctl(ctx);
}

// The page-locked, device-mapped words a launch and the host exchange (h_ctl: see host_ctx.h).
int ensure_ctl(mcalf_ctx* ctx) {
    if (!ctx->h_ctl) {
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_ctl, kCtlWords * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset((void*)ctx->h_ctl, 0, kCtlWords * sizeof(unsigned int));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->d_ctl, (void*)ctx->h_ctl, 0));
    }
    return MCALF_OK;
}

// Enqueue ONE streaming launch over `batch` live points (and the finalize kernel of a tiled spectrum) on `stream`.
// dP: the parameter rows as the DEVICE addresses them (HBM, or page-locked host memory); wgs: workgroups dedicated to
// the set-up while rows are outstanding (a multiple of the XCD count: so many per XCD); eager_rows: blocks of eight
// rows PER XCD that any of its workgroups may set up; staged: the kernel waits for the host's row count
// (ctx->h_ctl[kCtlArrived]).
int stream_launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, double* d_out, hipStream_t stream, int wgs,
                  int64_t eager_rows, bool staged, bool host_rows, bool from_cube) {
    // (from_cube: the rows are unit-cube rows, mapped through the prior box while they are decoded; the transformed rows
    // themselves, when the caller wants them, are formed on the host while the launch runs: host_scale_cube)
    KArgs a = make_kargs(ctx, mode, dP, 0, batch, 0, 0, 0, d_out, nullptr, from_cube, nullptr);
    a.taps_shared = 0;                                    // (a row's stamp covers its own taps only)
    a.recs = ctx->s_recs; a.taps = ctx->s_taps; a.hdr = ctx->s_hdr;
    a.persist = 1;
    a.order = nullptr;
    const int grid = 2 * ctx->num_cu;
    if (++ctx->stream_gen == 0u) {                        // stamps wrapped: none of the old ones may match again
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        HIP_TRY(ctx, hipMemset(ctx->d_ready, 0, ctx->cap_sws * sizeof(unsigned int)));
        HIP_TRY(ctx, hipDeviceSynchronize());
        ctx->stream_gen = 1u;
    }
    a.sctl = ctx->d_sctl;
    a.ready = ctx->d_ready;
    a.status = ctx->d_ctl;
    a.arrived = staged ? ctx->d_ctl + kCtlArrived : nullptr;
    a.gen = ctx->stream_gen;
    a.stream_wgs = wgs / kXcds;                           // (per XCD)
    // The words this launch will write are cleared first: the completion word of the previous launch must not be taken
    // for this one's, whatever the generation counts are.
    ctx->h_ctl[1] = 0u; ctx->h_ctl[2] = 0u; ctx->h_ctl[3] = 0u; ctx->h_ctl[kCtlFinalized] = 0u;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    a.eager_rows = (int)std::min<int64_t>((batch + 7) / 8, eager_rows);      // (local blocks of eight rows per XCD)
    a.spin_ticks = (long long)(ctx->stream_timeout_s * 1e8);
    a.rec_stride = (int)ctx->s_rec_stride; a.tap_stride = (int)ctx->s_tap_stride; a.hdr_stride = (int)ctx->s_hdr_stride;
    a.Pdev = host_rows ? ctx->s_P : nullptr;
    a.rest_chunk = ctx->stream_chunk;
    ctx->last.persistent = 1; ctx->last.grid = grid; ctx->last.items = a.nitems; ctx->last.lines_per_sync = ctx->lps;
    ctx->last.selfhalo = ctx->selfhalo; ctx->last.ordered = 0; ctx->last.inline_setup = 0;
    ctx->last.stream_setup_wgs = wgs;
    const bool timed = ctx->profiling && ctx->ev_used + 2 <= ctx->ev.size();
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[ctx->ev_used], stream));
    {
        void* kargs[] = {(void*)&a};
        HIP_TRY(ctx, hipLaunchKernel(fused_kernel_ptr(ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX, ctx->selfhalo != 0, ctx->lps, false, true),
                                     dim3((unsigned)grid), dim3(kBlock), kargs, ctx->lds_bytes, stream));
    }
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[ctx->ev_used + 1], stream));
        ctx->ev_used += 2;
    }
    // (tiled spectra: the finalize kernel writes the results; when they go to page-locked host memory it also tells the host)
    if ((mode == kModeLogL || mode == kModeChi2) && ctx->ntiles > 1) return launch_finalize(ctx, a, batch, mode, stream, 0, host_rows);
    return MCALF_OK;
}

bool stream_qualifies(const mcalf_ctx* ctx, int64_t batch) {
    const int64_t slots = 2LL * ctx->num_cu, nitems = batch * ctx->ntiles;
    // (never a wide-LSF context: its convolution runs in the wide kernels behind a convolution-free fused launch -- the
    // streaming workspaces and the tile's LDS carry no halo for it)
    return !ctx->wide && ctx->persist && nitems >= ctx->stream_min * slots && nitems <= 0x7fff0000LL && batch <= 0x7fff0000LL;
}

// Which XCDs do workgroups launched on the context's stream run on?  The streaming launch deals its rows to the XCDs
// 0 .. 7 of an unpartitioned MI355X and only a workgroup that RUNS on XCD x sets up and evaluates x's rows: on a DPX /
// QPX / CPX partition (one logical device = 4 / 2 / 1 XCDs), or on a stream / process restricted by a CU mask, some of
// the eight would see no workgroup and their rows would never be evaluated.  So the context asks the hardware -- a probe
// launch of eight workgroups per CU, each OR-ing 1 << XCC_ID into a word -- and takes the streaming launch only when the
// answer is exactly 0xFF (everything else: the row-block pipeline, run_host_pipelined).  run_host_stream additionally
// checks the arrival counts of every launch.
int stream_probe_xcds(mcalf_ctx* ctx) {
    ctx->xcd_mask = 0;
    unsigned int* d_mask = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_mask, sizeof(unsigned int)));
    hipError_t e = hipMemsetAsync(d_mask, 0, sizeof(unsigned int), ctx->stream);
    if (e == hipSuccess) {
        void* kargs[] = {(void*)&d_mask};
        e = hipLaunchKernel(xcd_probe_kernel_ptr(), dim3((unsigned)(8 * ctx->num_cu)), dim3(64), kargs, 0, ctx->stream);
    }
    unsigned int mask = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&mask, d_mask, sizeof mask, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_mask);
    if (e != hipSuccess) return set_err(ctx, MCALF_ERR_HIP, "XCD probe failed: %s", hipGetErrorString(e));
    ctx->xcd_mask = mask;
#ifdef MCALF_TESTING
    // (test builds only: make the context BELIEVE another answer, so that the check behind every launch can be exercised)
    if (ctx->env.test_xcd_mask >= 0) ctx->xcd_mask = (unsigned int)ctx->env.test_xcd_mask;              // MCALF_TEST_XCD_MASK
#endif
    return MCALF_OK;
}

extern "C" int mcalf_stream_partition(int32_t nrows, int32_t nxcd, int32_t* owner, int32_t* local) {
    if (nrows < 0 || nxcd < 1 || nxcd > 16 || (nrows > 0 && (!owner || !local)))
        return set_err(nullptr, MCALF_ERR_INVALID, "mcalf_stream_partition: nrows >= 0, 1 <= nxcd <= 16, non-NULL arrays");
    for (int32_t r = 0; r < nrows; ++r) owner[r] = local[r] = -1;
    for (int x = 0; x < nxcd; ++x) {
        const int n = stream_rows_of(nrows, x, nxcd);
        for (int j = 0; j < n; ++j) {
            const int r = stream_row(x, j, nxcd);
            if (r < 0 || r >= nrows || owner[r] != -1)
                return set_err(nullptr, MCALF_ERR_RANGE, "mcalf_stream_partition: row %d of XCD %d (local %d) is out of range or dealt twice", r, x, j);
            owner[r] = x; local[r] = j;
        }
    }
    return MCALF_OK;
}

// theta = cube * (hi - lo) + lo with the separately rounded multiply and add numpy performs (hires_fitter.py:206 / :214)
// and int() on the ncomp slot (:207-208): the arithmetic of sample_param() / mcalf_scale_cube_kernel, on the host -- the
// host-pointer cube entries form the rows they hand back while their launch runs.
void host_scale_cube(const mcalf_ctx* ctx, const double* cube, int64_t batch, double* theta) {
    const int nd = ctx->ndim;
    const double* lo = ctx->h_prior.data();
    const double* hi = lo + nd;
    for (int64_t r = 0; r < batch; ++r) {
        const double* c = cube + (size_t)r * nd;
        double* t = theta + (size_t)r * nd;
        for (int d = 0; d < nd; ++d) {
#pragma clang fp contract(off)
            const double scaled = c[d] * (hi[d] - lo[d]);
            t[d] = scaled + lo[d];
        }
        if (ctx->prior_int) t[ctx->startind] = std::trunc(t[ctx->startind]);
    }
}

// Large scalar-output batches through host pointers, the default plan: ONE streaming launch, no copy commands.
//
//   host                                         device (mcalf_fused_kernel<..., kStream = true>, the persistent grid)
//   launch the kernel                            every workgroup: set up eight of the first `eager_rows` live points
//   pageable P: copy the rows into the           (the rows the grid's first items need), reading the parameter rows
//     page-locked block, 128 at a time,          over PCIe from the page-locked block; then walk over work items.  The first
//     publishing the count after each            `stream_wgs` workgroups go on setting up the remaining rows, in ticket order,
//   (page-locked P: nothing to do)               as the host's count allows, and join the others at the item queue afterwards.
//   poll the word the last workgroup out         An item enters the component loop once its row's stamp is there; logL goes
//     writes; copy logL out if pageable          straight into page-locked memory; the last workgroup out re-arms the queues.
//
// Against the row-block pipeline (run_host_pipelined) this removes the copy commands, three of four set-up launches and
// the tails of the sub-threshold launches, and the GPU starts before a single row has been staged.  `*taken` = false when
// the call does not qualify (the caller then runs the pipeline); a wait that runs out inside the kernel (host thread
// stalled for longer than MCALF_STREAM_TIMEOUT) fails over to the pipeline too, after the grid has drained.
// On any device shape but the eight XCDs of an unpartitioned MI355X the call does not qualify either (stream_probe_xcds), and
// a launch whose kernel reports an XCD without workgroups is discarded like one that timed out.
int run_host_stream(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, double* out_scalar, bool* taken,
                    bool from_cube, double* theta_out) {
    *taken = false;
    const bool trace = ctx->stream_trace;
    double tm[7] = {};
    if (trace) tm[0] = now_us();
    ctx->last.stream_fallback = 0; ctx->last.stream_wgs_min = ctx->last.stream_wgs_max = 0;
    if (!ctx->stream_on || ctx->chunks_req > 0 || ctx->host_plan_n > 0 || ctx->profiling || !stream_qualifies(ctx, batch)) return MCALF_OK;
    // Tiled spectra (automatic): the streaming launch up to kStreamTiledMaxItems work items, the row-block pipeline beyond.
    // Measured on MI355X, config E (5 tiles per live point), pageable rows, host step over the device-resident one:
    // 2048 rows 1.11 streaming (1.24 pipeline), 4096 rows 1.075 (1.14), 8192 rows 1.057 (1.09), 16384 rows 1.051 (1.045) --
    // with the rows resident the tiled streaming kernel runs 4.3 % behind the batch kernel (single-tile: 3-4 % at 32768
    // rows: per-XCD queues and stamps, workgroups that keep setting rows up), the pipeline's cost is a fixed ~60-90 us of
    // block boundaries (profiles/r06_tiled_stream_crossover.txt).
    if (ctx->stream_on == 1 && ctx->ntiles > 1 && batch * (int64_t)ctx->ntiles > kStreamTiledMaxItems) return MCALF_OK;
    if (ctx->xcd_mask != (1u << kXcds) - 1u) {            // not the device shape the launch deals its rows for: see stream_probe_xcds
        ctx->last.stream_fallback = MCALF_STREAM_FALLBACK_SHAPE;
        return MCALF_OK;
    }
    const bool pin_in = is_pinned_host(P), pin_out = is_pinned_host(out_scalar);
    const double* dP_view = nullptr;
    if (pin_in && hipHostGetDevicePointer((void**)&dP_view, const_cast<double*>(P), 0) != hipSuccess) {
        (void)hipGetLastError();
        return MCALF_OK;                                  // page-locked but not device-mapped: the copy engines' job
    }
    double* d_out_view = nullptr;
    if (pin_out && hipHostGetDevicePointer((void**)&d_out_view, out_scalar, 0) != hipSuccess) {
        (void)hipGetLastError();
        return MCALF_OK;
    }
    int rc;
    if (trace) tm[1] = now_us();
    if ((rc = stream_prepare(ctx, mode, batch))) return rc;
    const size_t need = (pin_in ? 0 : (size_t)batch * rowlen) + (pin_out ? 0 : (size_t)batch);
    if (need > ctx->cap_stage) {
        if (ctx->h_stage) HIP_TRY(ctx, hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr; ctx->cap_stage = 0;
        // coherent: the kernel reads rows of this block while the host is still writing later ones
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_stage, need * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        ctx->cap_stage = need;
    }
    double* stage_in = pin_in ? nullptr : ctx->h_stage;
    double* stage_out = pin_out ? out_scalar : ctx->h_stage + (pin_in ? 0 : (size_t)batch * rowlen);
    if (!pin_in) HIP_TRY(ctx, hipHostGetDevicePointer((void**)&dP_view, stage_in, 0));
    if (!pin_out) HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_out_view, stage_out, 0));
    const int grid = 2 * ctx->num_cu;
    const int wgs = std::min((ctx->stream_wgs + kXcds - 1) / kXcds * kXcds, grid / 2 / kXcds * kXcds);
    // the rows the first items of an XCD's workgroups need: (workgroups per XCD - dedicated ones) tickets of its queue
    const int64_t first_rows = ((grid - wgs) / kXcds + ctx->ntiles - 1) / ctx->ntiles;
    ctx->h_ctl[0] = 0u;
    ctx->h_ctl[kCtlArrived] = 0u;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    ctx->last.path = MCALF_PATH_HOST_STREAM; ctx->last.row_blocks = 1;
    ctx->last.pinned_in = pin_in ? 1 : 0; ctx->last.pinned_out = pin_out ? 1 : 0;
    if (trace) tm[2] = now_us();
    // the rows an XCD's first items need are set up by whichever of its workgroups gets there first, the rest by its
    // dedicated workgroups
    const int64_t eager_blocks = ctx->stream_eager > 0 ? ctx->stream_eager : (first_rows + 7) / 8;
    if ((rc = stream_launch(ctx, mode, dP_view, batch, d_out_view, ctx->stream, wgs, eager_blocks, !pin_in, true, from_cube)))
        return rc;
    const bool tiled = (mode == kModeLogL || mode == kModeChi2) && ctx->ntiles > 1;
    if (trace) tm[3] = now_us();
    // The kernel is in flight (or about to be): stage the rows.  Nothing below can fail before every row has been
    // published, so the grid never waits for a row that is not coming.
    if (!pin_in) {
        constexpr int64_t kRowsPerStep = 128;
        for (int64_t r0 = 0; r0 < batch; r0 += kRowsPerStep) {
            const int64_t n = std::min(kRowsPerStep, batch - r0);
            std::memcpy(stage_in + (size_t)r0 * rowlen, P + (size_t)r0 * rowlen, (size_t)n * rowlen * sizeof(double));
            __atomic_store_n(const_cast<unsigned int*>(ctx->h_ctl + kCtlArrived), (unsigned int)(r0 + n), __ATOMIC_RELEASE);
        }
    }
    if (theta_out) host_scale_cube(ctx, P, batch, theta_out);   // (under the launch)
    hipError_t he = hipGetLastError();
    if (trace) tm[4] = now_us();
    // Completion: the word the last workgroup writes once every result has been acknowledged (a stream wait costs an
    // interrupt and a thread wake-up); the stream is asked now and then so that a faulted launch cannot keep us here.
    bool polled = false;
    if (he == hipSuccess && ctx->stream_poll) {
        const unsigned int want = ctx->stream_gen;
        const volatile unsigned int* word = ctx->h_ctl + (tiled ? kCtlFinalized : 1);     // (tiled: the finalize kernel's word)
        for (unsigned long spins = 0;; ++spins) {
            if (__atomic_load_n(const_cast<unsigned int*>(word), __ATOMIC_ACQUIRE) == want) { polled = true; break; }
            if ((spins & 0xFFFFul) == 0xFFFFul && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;
            __builtin_ia32_pause();
        }
    }
    if (!polled) {
        const hipError_t se = hipStreamSynchronize(ctx->stream);
        if (he == hipSuccess) he = se;
    }
    ctx->last.stream_polled = polled ? 1 : 0;
    if (he != hipSuccess) return set_err(ctx, MCALF_ERR_HIP, "streaming launch failed: %s", hipGetErrorString(he));
    // Not an answer, although the grid has drained: a wave ran out of patience (h_ctl[0]), or an XCD received no workgroup
    // of the launch (h_ctl[2], the fewest arrivals: every XCD is dealt rows at the sizes that stream, and only its own
    // workgroups evaluate them -- the probe said eight XCDs, this launch met fewer).  The caller goes the pipelined way.
    const unsigned int gave_up = ctx->h_ctl[0];
    unsigned int fewest = ctx->h_ctl[2];
#ifdef MCALF_TESTING
    // (test builds only: pretend the kernel reported an XCD without workgroups -- no CU mask can produce one on an
    // unpartitioned device, see mcalf_set_cu_mask -- so that the path behind the report runs in a test)
    if (ctx->env.test_starve > 0) fewest = 0u;                                                          // MCALF_TEST_STARVE
#endif
    ctx->last.stream_wgs_min = (int32_t)fewest; ctx->last.stream_wgs_max = (int32_t)ctx->h_ctl[3];
    if (gave_up != 0u || fewest == 0u) {
        (void)hipStreamSynchronize(ctx->stream);
        HIP_TRY(ctx, hipMemset(ctx->d_sctl, 0, sizeof(StreamCtl)));
        HIP_TRY(ctx, hipDeviceSynchronize());
        ctx->last.stream_fallback = gave_up != 0u ? MCALF_STREAM_FALLBACK_TIMEOUT : MCALF_STREAM_FALLBACK_STARVED;
        if (gave_up == 0u) ctx->xcd_mask = 0;            // (what the probe said does not hold on this stream: no further attempts)
        return MCALF_OK;
    }
    if (trace) tm[5] = now_us();
    if (!pin_out) std::memcpy(out_scalar, stage_out, (size_t)batch * sizeof(double));
    if (trace) {
        tm[6] = now_us();
        for (int k = 0; k < 6; ++k) ctx->strace.t[k] += tm[k + 1] - tm[k];
        ctx->strace.n++;
    }
    *taken = true;
    return MCALF_OK;
}
