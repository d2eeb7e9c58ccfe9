// libmcalf_hip.so, host side: contexts and the C ABI of include/mcalf_hip.h -- creation, launches of the batch kernels,
// the device-pointer and host-pointer entries, prior transform, diagnostics.  (The streaming launch of large host-pointer
// batches is host_stream.cpp, the resident evaluator and the likelihood broker broker.cpp, the RCCL gather comm.cpp; the
// kernels are kernels.hip.)  There is no CPU fallback: every entry point fails with MCALF_ERR_NODEVICE when no gfx950
// device is present.
#include <cmath>
#include <new>

#include "host_ctx.h"

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

int set_err(mcalf_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

static int pick_device(mcalf_ctx* ctx, int requested, int* out_dev, std::string* arch) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return set_err(ctx, MCALF_ERR_NODEVICE, "no HIP device available (%s); libmcalf_hip has no CPU fallback",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    int dev = requested;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= count) return set_err(ctx, MCALF_ERR_INVALID, "device %d out of range (count %d)", dev, count);
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(ctx, MCALF_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", dev,
                       prop.gcnArchName);
    *out_dev = dev;
    if (arch) *arch = prop.gcnArchName;
    return MCALF_OK;
}

static int upload_tables(mcalf_ctx* ctx, double** d_tabs) {
    HIP_TRY(ctx, hipMalloc((void**)d_tabs, sizeof(VT_T_HOST)));
    HIP_TRY(ctx, hipMemcpy(*d_tabs, VT_T_HOST, sizeof(VT_T_HOST), hipMemcpyHostToDevice));
    return MCALF_OK;
}

#ifndef MCALF_SRC_HASH
#define MCALF_SRC_HASH "unstamped"      // mc-alf_amd/build.py passes the sha256 of the kernel sources
#endif
extern "C" const char* mcalf_version(void) { return "mcalf_hip 0.4 (gfx950, abi " MCALF_STR(MCALF_ABI_VERSION) ") src " MCALF_SRC_HASH; }

extern "C" const char* mcalf_last_error(const mcalf_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

extern "C" void mcalf_destroy(mcalf_ctx* ctx) {
    if (!ctx) return;
    if (is_multi(ctx) || ctx->pool) {                     // a multi-device parent owns its sub-contexts and workers, nothing else
        multi_release(ctx);
        delete ctx;
        return;
    }
    stream_trace_report(ctx);
    host_trace_report(ctx);
    for (auto& w : ctx->stagers) w->stop();
    ctx->stagers.clear();
    (void)hipSetDevice(ctx->device);
    comm_release(ctx);
    resident_stop(ctx);
    if (ctx->res_stream) (void)hipStreamDestroy(ctx->res_stream);
    if (ctx->h_box) (void)hipHostFree((void*)ctx->h_box);
    if (ctx->d_res_shared) (void)hipFree(ctx->d_res_shared);
    void* bufs[] = {ctx->d_nu, ctx->d_obj, ctx->d_ispec2, ctx->d_lgis, ctx->d_err, ctx->d_tabs, ctx->d_lines, ctx->d_wtab, ctx->d_segok,
                    ctx->d_P,  ctx->d_out, ctx->d_partial, ctx->d_model, ctx->d_bounds, ctx->d_prior, ctx->d_recs, ctx->d_taps, ctx->d_hdr,
                    ctx->d_queue, ctx->d_order, ctx->d_sws, ctx->d_wide, ctx->d_wtaps, ctx->d_wpartial, ctx->d_wrows, ctx->d_whdr};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->h_ctl) (void)hipHostFree((void*)ctx->h_ctl);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (hipEvent_t e : ctx->ev_h2d)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_join)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : ctx->aux)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->ev_kernels) (void)hipEventDestroy(ctx->ev_kernels);
    for (hipEvent_t e : ctx->ev_comm)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int create_impl(const mcalf_spec* sp, mcalf_ctx* ctx) {
    if (!sp) return set_err(ctx, MCALF_ERR_INVALID, "spec is NULL");
    if (sp->npix <= 0 || !sp->wl || !sp->flux || !sp->err)
        return set_err(ctx, MCALF_ERR_INVALID, "npix must be > 0 and wl/flux/err non-NULL");
    if (sp->npix > (1 << 30)) return set_err(ctx, MCALF_ERR_RANGE, "npix too large");
    if (sp->nlines <= 0 || !sp->lines) return set_err(ctx, MCALF_ERR_INVALID, "need at least one line");
    if (sp->ncompmax < 0 || sp->nfill < 0) return set_err(ctx, MCALF_ERR_INVALID, "negative component counts");
    if (!(sp->velstep > 0.0)) return set_err(ctx, MCALF_ERR_INVALID, "velstep must be > 0");
    if (sp->conv_mode != MCALF_CONV_WRAP_NUMPY && sp->conv_mode != MCALF_CONV_SAME_EDGE_JAX)
        return set_err(ctx, MCALF_ERR_INVALID, "unknown conv_mode %d", sp->conv_mode);
    double rmax = sp->specres_max;
    if (!sp->freespecres && !(rmax >= sp->specres_fixed)) rmax = sp->specres_fixed;
    if (!(rmax > 0.0) || !std::isfinite(rmax))
        return set_err(ctx, MCALF_ERR_INVALID, "specres_max must be finite and > 0");

    int rc = pick_device(ctx, sp->device, &ctx->device, &ctx->arch);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    ctx->npix = sp->npix;
    ctx->nlines = sp->nlines;
    ctx->ncompmax = sp->ncompmax;
    ctx->nfill = sp->nfill;
    ctx->freespecres = sp->freespecres ? 1 : 0;
    ctx->freecont = sp->freecont ? 1 : 0;
    ctx->conv_mode = sp->conv_mode;
    ctx->specres_fixed = sp->specres_fixed;
    ctx->specres_max = rmax;
    ctx->contval_fixed = sp->contval_fixed;
    ctx->velstep = sp->velstep;
    ctx->asymm = sp->asymmlike ? 1 : 0;                                  // hires_fitter.py:296-303
    ctx->veto4 = sp->asymm_n4 + 0.01 * (double)sp->npix;                 // gauss_cdf[1] + gracenum (:181,302)
    ctx->veto5 = sp->asymm_n5 + 0.01 * (double)sp->npix;                 // gauss_cdf[2] + gracenum (:300)
    ctx->startind = ctx->freecont + ctx->freespecres;             // hires_fitter.py:169-174
    ctx->endind = ctx->startind + 3 * ctx->ncompmax + 1;          // :176
    ctx->ndim = ctx->endind + 3 * ctx->nfill;                     // :184-200

    // LSF reach and tiling
    const double sigma_max = (rmax / kFwhmToSigma) / sp->velstep;
    if (ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX) {
        ctx->jax_half = (int)std::ceil((float)(kKernelReach * sigma_max));   // :557-559 (float32 ceil)
        ctx->n_cap = ctx->jax_half;
        if (2L * ctx->jax_half + 1 > ctx->npix)
            return set_err(ctx, MCALF_ERR_INVALID,
                           "JAX-path LSF kernel (%d taps) is longer than the spectrum (%ld px): the reference's "
                           "jnp.convolve(..., 'same') / jnp.where (hires_fitter.py:674-681) cannot broadcast either",
                           2 * ctx->jax_half + 1, ctx->npix);
    } else {
        ctx->n_cap = (rmax > sp->velstep) ? (int)std::ceil(kKernelReach * sigma_max) : 0;
    }
    ctx->ncl_cap = std::max(1, ctx->ncompmax * ctx->nlines + ctx->nfill);
    // Lines per barrier: 5 when that saves a barrier at the context's largest line count and the extra folded
    // tables cost no tile pixels (LDS), else 4.
    auto fixed_for = [&](int lps) {
        return 2 * (size_t)lps * kTabPad + (size_t)ctx->ncl_cap * kRecStride + (2 * (size_t)ctx->n_cap + 8) + kRedDoubles +
               64 * VT_INODES;
    };
    auto ext_for = [&](int lps) {
        size_t e = kExtMax;
        while (e > 0 && (fixed_for(lps) + tile_doubles((int)e)) * sizeof(double) > kLdsBudget) e -= 64;
        return e;
    };
    ctx->lps = ((ctx->ncl_cap + 4) / 5 < (ctx->ncl_cap + 3) / 4 && ext_for(5) == ext_for(4)) ? 5 : 4;
    if (ctx->env.lines_per_sync == 4 || (ctx->env.lines_per_sync == 5 && ext_for(5) == ext_for(4))) ctx->lps = ctx->env.lines_per_sync;
    if (ext_for(ctx->lps) < 2 * (size_t)ctx->n_cap + 64) {
        // The LSF does not fit a workgroup tile with its halo: the fused kernel then runs without convolution (a tile
        // without halo) and the wide kernels convolve behind it (launch_wide) -- the reference simply builds a longer
        // kernel (hires_fitter.py:458-464; JAX semantics: its fixed grid, :549-560, whatever its length).
        if (2.0 * (double)ctx->n_cap + 1.0 > 6.0e7)
            return set_err(ctx, MCALF_ERR_RANGE,
                           "LSF half-width %d px (specres_max %.3g km/s at %.3g km/s/px) with %d component-lines does not "
                           "fit a %d-pixel workgroup tile / the %zu-byte LDS budget", ctx->n_cap, rmax, sp->velstep,
                           ctx->ncl_cap, kExtMax, kLdsBudget);
        ctx->wide = 1;
        ctx->wide_n_cap = ctx->n_cap;
        ctx->n_cap = 0;
        ctx->lps = ((ctx->ncl_cap + 4) / 5 < (ctx->ncl_cap + 3) / 4 && ext_for(5) == ext_for(4)) ? 5 : 4;
    }
    const size_t fixed_doubles = fixed_for(ctx->lps);
    size_t ext = ext_for(ctx->lps);
    if (ext < 2 * (size_t)ctx->n_cap + 64)
        return set_err(ctx, MCALF_ERR_RANGE, "%d component-lines do not fit the %zu-byte LDS budget", ctx->ncl_cap, kLdsBudget);
    // tiles are multiples of 8 pixels (the epilogue works in aligned groups of 8), balanced over the spectrum
    const long tmax = ((long)ext - 2 * ctx->n_cap) & ~7L;
    long tile = std::min(tmax, (ctx->npix + 7) & ~7L);
    long ntiles = (ctx->npix + tile - 1) / tile;
    tile = (((ctx->npix + ntiles - 1) / ntiles) + 7) & ~7L;
    ntiles = (ctx->npix + tile - 1) / tile;
    ctx->tile = (int)tile;
    ctx->ntiles = (int)ntiles;
    ctx->lds_bytes = (fixed_doubles + (size_t)tile_doubles((int)tile + 2 * ctx->n_cap)) * sizeof(double);
    ctx->lds_bytes_inline = (fixed_for(4) + (size_t)tile_doubles((int)tile + 2 * ctx->n_cap)) * sizeof(double);
    ctx->selfhalo = (ntiles == 1 && ctx->n_cap < ctx->npix) ? 1 : 0;

    // spectrum arrays (float64 host arithmetic identical to the reference's numpy expressions)
    // Self-halo contexts index nu by thread (0 .. kExtMax-1): the entries past the spectrum are VIRTUAL pixels.
    // Up to the next multiple of 64 they continue the wavelength grid when it is recognisably linear or
    // logarithmic over its last 64 pixels (so that the last, partial segment can be interpolated like the
    // others; the virtual pixels' own results are never stored); beyond that they repeat the last value.
    const long nu_len = ctx->selfhalo ? (long)kExtMax : ctx->npix;
    std::vector<double> nu(nu_len), is2(ctx->npix), lg(ctx->npix);
    for (long i = 0; i < ctx->npix; ++i) {
        const double wave_cm = sp->wl[i] / 1e8;            // :376
        nu[i] = kCcgs / wave_cm;                           // :362 at zp1 = 1
        is2[i] = 1.0 / (sp->err[i] * sp->err[i]);          // :292
        lg[i] = std::log(is2[i]);                          // :294
    }
    if (ctx->selfhalo) {
        const long n = ctx->npix, upto = std::min<long>(nu_len, (n + 63) & ~63L);
        int kind = 0;                                      // 1 linear, 2 logarithmic
        if (n >= 66) {
            const double d = sp->wl[n - 1] - sp->wl[n - 2], r = sp->wl[n - 1] / sp->wl[n - 2];
            bool lin = true, lg_ = true;
            for (long i = n - 64; i < n - 1; ++i) {
                if (!(std::fabs((sp->wl[i + 1] - sp->wl[i]) - d) <= 1e-9 * std::fabs(d))) lin = false;
                if (!(std::fabs(sp->wl[i + 1] / sp->wl[i] - r) <= 1e-9 * std::fabs(r - 1.0))) lg_ = false;
            }
            kind = lin ? 1 : (lg_ ? 2 : 0);
        }
        // the common step from the pixels 64 apart (averages the grid's own rounding noise)
        const double step = kind == 1 ? (sp->wl[n - 1] - sp->wl[n - 65]) / 64.0
                          : kind == 2 ? std::exp(std::log(sp->wl[n - 1] / sp->wl[n - 65]) / 64.0) : 0.0;
        for (long i = n; i < nu_len; ++i) {
            if (kind != 0 && i < upto) {
                const double m = (double)(i - (n - 1));
                const double wl = kind == 1 ? sp->wl[n - 1] + m * step : sp->wl[n - 1] * std::pow(step, m);
                nu[i] = kCcgs / (wl / 1e8);
            } else {
                // past the last (partial) segment nothing is ever stored: nu = 0 puts these lanes at u = -nu0/dnu,
                // thousands of Doppler widths away from every line, so their segments go the cheap node-only way
                nu[i] = (i >= upto) ? 0.0 : nu[i - 1];
            }
        }
    }
    std::vector<LineDev> lines(ctx->nlines + 1);
    for (int l = 0; l <= ctx->nlines; ++l) {
        const mcalf_line& src = (l < ctx->nlines) ? sp->lines[l] : sp->fill;
        lines[l].wrest_cm = src.wrest_A / 1e8;             // :376
        lines[l].f = src.f;
        lines[l].gamma4pi = src.gamma / (4.0 * M_PI);      // :361
        lines[l].nujk = kCcgs / lines[l].wrest_cm;         // :359
    }
    const size_t nb = (size_t)ctx->npix * sizeof(double);
    const size_t nbp = nb + 8 * sizeof(double);          // the epilogue reads 8 pixels per thread without a bounds test
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_nu, (size_t)nu_len * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_obj, nbp));
    HIP_TRY(ctx, hipMemset(ctx->d_obj, 0, nbp));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ispec2, nbp));
    HIP_TRY(ctx, hipMemset(ctx->d_ispec2, 0, nbp));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_lgis, nbp));
    HIP_TRY(ctx, hipMemset(ctx->d_lgis, 0, nbp));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_err, nbp));
    HIP_TRY(ctx, hipMemset(ctx->d_err, 0, nbp));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_lines, lines.size() * sizeof(LineDev)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_nu, nu.data(), (size_t)nu_len * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_obj, sp->flux, nb, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_ispec2, is2.data(), nb, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_lgis, lg.data(), nb, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_err, sp->err, nb, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_lines, lines.data(), lines.size() * sizeof(LineDev), hipMemcpyHostToDevice));
    rc = upload_tables(ctx, &ctx->d_tabs);
    if (rc) return rc;

    // Far-wing interpolation set-up: which 64-pixel segments of each tile may be interpolated in
    // pixel-index space.  A segment qualifies when it lies inside the tile's extent without crossing
    // the periodic seam and nu(pixel) itself is reproduced by the 8-node interpolant to 1e-15 (true
    // for linear / logarithmic wavelength grids, false across masked gaps).
    std::vector<unsigned long long> segok(ctx->ntiles, 0ULL);
    double dnu_seg = 0.0;
    for (int t = 0; t < ctx->ntiles; ++t) {
        const long t0 = (long)t * ctx->tile;
        const long tlen = std::min<long>(ctx->tile, ctx->npix - t0);
        const long ext0 = ctx->selfhalo ? 0 : t0 - ctx->n_cap, extCount = tlen + 2L * ctx->n_cap;
        for (int m = 0; m < 64; ++m) {
            const long i0 = 64L * m;
            const long e0 = ext0 + i0;
            if (ctx->selfhalo) {
                if (e0 + 63 >= nu_len) continue;
                if (e0 >= ((ctx->npix + 63) & ~63L)) {          // wholly virtual (nu = 0): nothing to get wrong
                    segok[t] |= 1ULL << m;
                    continue;
                }
                // otherwise: real pixels, the last segment completed by the virtual continuation of the grid
            } else {
                if (i0 + 64 > extCount) continue;
                if (e0 < 0 || e0 + 63 >= ctx->npix) continue;
            }
            bool ok = true;
            const double dir = nu[e0 + 63] - nu[e0];
            for (int i = 0; i < 64 && ok; ++i) {
                double v = 0.0;
                for (int k = 0; k < VT_INODES; ++k) v += VT_INTERP_W_HOST[i * VT_INODES + k] * nu[e0 + VT_INTERP_NODES[k]];
                const double x = nu[e0 + i];
                if (!std::isfinite(x) || !(std::fabs(v - x) <= 1e-15 * std::fabs(x))) ok = false;
                if (i > 0 && !((nu[e0 + i] - nu[e0 + i - 1]) * dir > 0.0)) ok = false;   // strictly monotonic
            }
            if (!ok) continue;
            segok[t] |= 1ULL << m;
            dnu_seg = std::max(dnu_seg, std::fabs(dir));
        }
    }
    ctx->dnu_seg = dnu_seg;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_segok, segok.size() * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_segok, segok.data(), segok.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_wtab, sizeof(VT_INTERP_W_HOST)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_wtab, VT_INTERP_W_HOST, sizeof(VT_INTERP_W_HOST), hipMemcpyHostToDevice));
    // more than 64 KiB of dynamic LDS needs the attribute (2 workgroups x 78 KiB fit the 160 KiB of a CU)
    for (int k = 0; k < fused_kernel_count(); ++k)
        HIP_TRY(ctx, hipFuncSetAttribute(fused_kernel_at(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
    if ((rc = create_stream(ctx, &ctx->stream))) return rc;
    {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if ((rc = stream_probe_xcds(ctx))) return rc;         // (the streaming launch is for one device shape only)
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_queue, kMaxChunks * sizeof(unsigned int)));
        HIP_TRY(ctx, hipMemset(ctx->d_queue, 0, kMaxChunks * sizeof(unsigned int)));
        // (the environment's snapshot was applied at the top of mcalf_create; what depends on the device is finished here)
        ctx->inline_max_items = ctx->env.inline_max >= 0 ? ctx->env.inline_max : 2 * ctx->num_cu;   // launches that fit the chip in one round of workgroups
        if (ctx->env.stream_wgs >= 0) ctx->stream_wgs = std::min(ctx->env.stream_wgs, ctx->num_cu);
    }
    return MCALF_OK;
}

extern "C" int mcalf_create(const mcalf_spec* spec, mcalf_ctx** out) {
    if (!out) return set_err(nullptr, MCALF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    mcalf_ctx* ctx = new (std::nothrow) mcalf_ctx();
    if (!ctx) return set_err(nullptr, MCALF_ERR_NOMEM, "out of host memory");
    ctx->env = read_environment();                        // ONE snapshot per context; nothing else reads the environment
    apply_environment(ctx);
    int rc = create_impl(spec, ctx);
    if (rc != MCALF_OK) {
        g_last_error = ctx->err;
        mcalf_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return MCALF_OK;
}

extern "C" int mcalf_info(const mcalf_ctx* ctx, mcalf_info_t* info) {
    if (!ctx || !info) return set_err(nullptr, MCALF_ERR_INVALID, "NULL argument");
    memset(info, 0, sizeof *info);
    info->abi_version = MCALF_ABI_VERSION;
    info->ndim = ctx->ndim;
    info->startind = ctx->startind;
    info->endind = ctx->endind;
    info->n_cap = ctx->wide ? ctx->wide_n_cap : ctx->n_cap;      // (what specres_max provisions, wherever the convolution runs)
    info->tile = ctx->tile;
    info->ntiles = ctx->ntiles;
    info->device = ctx->device;
    info->npix = ctx->npix;
    snprintf(info->arch, sizeof info->arch, "%s", ctx->arch.c_str());
    info->ndevices = is_multi(ctx) ? (int32_t)ctx->subs.size() : 1;
    for (int k = 0; k < info->ndevices && k < 16; ++k) info->devices[k] = is_multi(ctx) ? ctx->subs[k]->device : ctx->device;
    return MCALF_OK;
}

extern "C" int mcalf_last_launch(const mcalf_ctx* ctx, mcalf_launch_info_t* info) {
    if (!ctx || !info) return set_err(nullptr, MCALF_ERR_INVALID, "NULL argument");
    if (is_multi(ctx)) {                                  // sub-context 0's launch, and how many devices the call was cut over
        const int rc = mcalf_last_launch(ctx->subs[0], info);
        info->devices_used = ctx->multi_last_active;
        return rc;
    }
    *info = ctx->last;
    info->xcd_mask = (int32_t)ctx->xcd_mask;
    info->devices_used = 1;
    return MCALF_OK;
}

// A stream of the context: non-blocking, or -- with a CU mask (mcalf_set_cu_mask) -- restricted to the mask's CUs.
// priority: +1 the copy stream, -1 the second compute stream of the row-block pipeline, 0 everything else.  A process has
// few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and HIP deals its streams to them per PRIORITY level: in a process
// with many streams (an application with several contexts; bench.py's legs next to its main context) two of the pipeline's
// three streams can land on ONE queue -- its row blocks then run strictly one after the other (measured with
// MCALF_HOST_TRACE=2: config E's host step 2.90 -> 2.99 ms), or its copies wait behind kernels again.  Streams of different
// priority levels take their queues from different pools, so the three never share.
int create_stream(mcalf_ctx* ctx, hipStream_t* out, int priority) {
    if (!ctx->cu_mask.empty()) {
        HIP_TRY(ctx, hipExtStreamCreateWithCUMask(out, (uint32_t)ctx->cu_mask.size(), ctx->cu_mask.data()));
        return MCALF_OK;
    }
    int least = 0, greatest = 0;
    if (priority != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        HIP_TRY(ctx, hipStreamCreateWithPriority(out, hipStreamNonBlocking, priority > 0 ? greatest : least));
        return MCALF_OK;
    }
    (void)hipGetLastError();
    HIP_TRY(ctx, hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return MCALF_OK;
}

extern "C" int mcalf_set_cu_mask(mcalf_ctx* ctx, const uint32_t* mask, int32_t nwords) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (nwords < 0 || nwords > 64 || (nwords > 0 && !mask)) return set_err(ctx, MCALF_ERR_INVALID, "CU mask: 0 .. 64 words");
    bool any = nwords == 0;
    for (int32_t i = 0; i < nwords; ++i) any = any || mask[i] != 0u;
    if (!any) return set_err(ctx, MCALF_ERR_INVALID, "CU mask selects no compute unit");
    if (is_multi(ctx)) {
        for (mcalf_ctx* sub : ctx->subs) {
            const int rc = mcalf_set_cu_mask(sub, mask, nwords);
            if (rc) return set_err(ctx, rc, "%s", sub->err.c_str());
        }
        return MCALF_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // EVERY stream the context owns is replaced: the launch stream and its auxiliaries, the resident evaluator's (its
    // kernel is told to leave first: the next one-theta call starts another one on the new stream) and the exchange
    // stream of the library's gather (pending exchanges are waited for; it is re-created by the next gather).
    resident_stop(ctx);
    if (ctx->res_stream) { HIP_TRY(ctx, hipStreamSynchronize(ctx->res_stream)); HIP_TRY(ctx, hipStreamDestroy(ctx->res_stream)); ctx->res_stream = nullptr; }
    if (ctx->comm_stream) { HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream)); HIP_TRY(ctx, hipStreamDestroy(ctx->comm_stream)); ctx->comm_stream = nullptr; }
    // the context's own streams are idle between its (synchronous) host-pointer calls; wait anyway, then replace them
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t& st : ctx->aux)
        if (st) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipStreamDestroy(st)); st = nullptr; }
    HIP_TRY(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = nullptr;
    ctx->cu_mask.assign(mask, mask + nwords);
    int rc = create_stream(ctx, &ctx->stream);
    if (rc != MCALF_OK) {                                 // (a mask the runtime refuses: back to an unrestricted stream)
        ctx->cu_mask.clear();
        const std::string why = ctx->err;
        if (create_stream(ctx, &ctx->stream) != MCALF_OK) return rc;
        (void)stream_probe_xcds(ctx);
        return set_err(ctx, rc, "%s (the context keeps an unrestricted stream)", why.c_str());
    }
    return stream_probe_xcds(ctx);
}

// Live points per pass of a wide-LSF launch (launch_wide): each of its scratch buffers -- unconvolved spectra, taps -- stays
// below 512 MB; 0 when a single live point does not fit.
static int64_t wide_rows_per_pass(const mcalf_ctx* ctx, int64_t batch) {
    constexpr size_t kScratchBytes = (size_t)512 << 20;
    const size_t per_row = sizeof(double) * (size_t)std::max<int64_t>(ctx->npix, 2 * (int64_t)ctx->wide_n_cap + 1);
    if (per_row > kScratchBytes) return 0;
    return std::min<int64_t>(batch, std::min<int64_t>(65535, std::max<int64_t>(1, (int64_t)(kScratchBytes / per_row))));
}

static int grow_sample_ws(mcalf_ctx* ctx, int64_t batch) {
    int rc;
    if ((rc = grow(ctx, &ctx->d_recs, &ctx->cap_recs, (size_t)batch * ctx->ncl_cap * kRecStride))) return rc;
    if ((rc = grow(ctx, &ctx->d_taps, &ctx->cap_taps, (size_t)batch * (2 * (size_t)ctx->n_cap + 8)))) return rc;
    if ((rc = grow(ctx, &ctx->d_hdr, &ctx->cap_hdr, (size_t)batch))) return rc;
    if ((rc = grow(ctx, &ctx->d_order, &ctx->cap_order, (size_t)batch))) return rc;
    return MCALF_OK;
}

extern "C" int mcalf_reserve(mcalf_ctx* ctx, int64_t batch) {
    if (!ctx || batch < 0) return set_err(ctx, MCALF_ERR_INVALID, "bad arguments");
    if (is_multi(ctx)) {                                  // every device's largest shard of such a batch
        const int n = multi_active(ctx, batch);
        for (int k = 0; k < n; ++k) {
            const int rc = mcalf_reserve(ctx->subs[k], (batch + n - 1) / n);
            if (rc) return set_err(ctx, rc, "%s", ctx->subs[k]->err.c_str());
        }
        return MCALF_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = grow(ctx, &ctx->d_P, &ctx->cap_P, (size_t)batch * std::max(ctx->ndim, 5)))) return rc;
    if ((rc = grow(ctx, &ctx->d_out, &ctx->cap_out, (size_t)batch))) return rc;
    if ((rc = grow(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)batch * ctx->ntiles * 4))) return rc;
    if ((rc = grow_sample_ws(ctx, batch))) return rc;
    if (ctx->wide && batch > 0) {                         // the scratch of a wide-LSF launch (one pass of launch_wide)
        const int64_t rows = wide_rows_per_pass(ctx, batch);
        if (rows < 1) return set_err(ctx, MCALF_ERR_RANGE, "wide LSF: one live point exceeds the scratch of a pass");
        const int nblocks = (int)((ctx->npix + kWideBlockPix - 1) / kWideBlockPix);
        if ((rc = grow(ctx, &ctx->d_wide, &ctx->cap_wide, (size_t)rows * ctx->npix))) return rc;
        if ((rc = grow(ctx, &ctx->d_wtaps, &ctx->cap_wtaps, (size_t)rows * (2 * (size_t)ctx->wide_n_cap + 1)))) return rc;
        if ((rc = grow(ctx, &ctx->d_whdr, &ctx->cap_whdr, (size_t)rows))) return rc;
        if ((rc = grow(ctx, &ctx->d_wpartial, &ctx->cap_wpartial, (size_t)rows * nblocks * 4))) return rc;
        if ((rc = grow(ctx, &ctx->d_wrows, &ctx->cap_wrows, (size_t)rows * 5))) return rc;
    }
    return MCALF_OK;
}

// Enqueue rows [row0, row0 + nrows) of a batch on `stream`: set-up kernel, fused kernel, finalize when tiled.
// `chunk` selects the row of the shared tap table this block writes and reads (fixed-resolution contexts).
// `from_cube`: dP holds unit-cube rows, mapped through the prior box while decoding; d_theta (optional)
// receives the transformed rows.
// The kernel arguments of rows [row0, row0 + nrows) of a batch (everything but the launch geometry).
KArgs make_kargs(const mcalf_ctx* ctx, int mode, const double* dP, int64_t row0, int64_t nrows, int chunk,
                        int targonly, int onecomp_fill, double* d_out, double* d_model, bool from_cube, double* d_theta, int wide_stage) {
    const int rowlen = (mode == kModeOneComp) ? 5 : ctx->ndim;
    const size_t tapTotal = 2 * (size_t)ctx->n_cap + 8;
    KArgs a = {};
    a.taps_shared = (!ctx->freespecres && mode != kModeOneComp) ? 1 : 0;
    a.recs = ctx->d_recs + (size_t)row0 * ctx->ncl_cap * kRecStride;
    a.taps = ctx->d_taps + (a.taps_shared ? (size_t)chunk : (size_t)row0) * tapTotal;
    a.hdr = ctx->d_hdr + row0;
    a.nu = ctx->d_nu; a.obj = ctx->d_obj; a.ispec2 = ctx->d_ispec2; a.lgis = ctx->d_lgis; a.err = ctx->d_err;
    a.asymm = (mode == kModeLogL) ? ctx->asymm : 0; a.veto4 = ctx->veto4; a.veto5 = ctx->veto5;
    a.P = dP + (size_t)row0 * rowlen;
    a.partial = ctx->d_partial ? ctx->d_partial + (size_t)row0 * ctx->ntiles * 4 : nullptr;
    a.out = d_out ? d_out + row0 : nullptr;
    a.model = d_model ? d_model + (size_t)row0 * ctx->npix : nullptr;
    a.lines = ctx->d_lines; a.tabs = ctx->d_tabs; a.wtab = ctx->d_wtab; a.segok = ctx->d_segok; a.dnu_seg = ctx->dnu_seg;
    a.npix = (int)ctx->npix; a.ndim = ctx->ndim; a.ntiles = ctx->ntiles; a.tile = ctx->tile;
    a.n_cap = ctx->n_cap; a.ncl_cap = ctx->ncl_cap;
    a.nlines = ctx->nlines; a.ncompmax = ctx->ncompmax; a.nfill = ctx->nfill;
    a.startind = ctx->startind; a.endind = ctx->endind;
    a.freespecres = ctx->freespecres; a.freecont = ctx->freecont;
    a.targonly = targonly; a.mode = mode; a.jax_half = ctx->jax_half; a.onecomp_fill = onecomp_fill;
    a.selfhalo = ctx->selfhalo;
    a.specres_fixed = ctx->specres_fixed; a.contval_fixed = ctx->contval_fixed; a.velstep = ctx->velstep;
    a.log2pi = std::log(2.0 * M_PI);
    a.prior_lo = from_cube ? ctx->d_prior : nullptr;
    a.prior_hi = from_cube ? ctx->d_prior + ctx->ndim : nullptr;
    a.theta_out = (from_cube && d_theta) ? d_theta + (size_t)row0 * ctx->ndim : nullptr;
    a.prior_int = ctx->prior_int;
    a.nitems = (int)(nrows * ctx->ntiles);
    a.nrows = (int)nrows;
    a.queue = ctx->d_queue + chunk;
    if (wide_stage == kWideFused) {
        // the convolution-free fused launch of a wide-LSF context: no resolution exceeds this step (hires_fitter.py:445), the
        // continuum is 1 -- the wide kernels apply both afterwards (the layout fields startind / endind stay the context's)
        // (JAX semantics convolve unconditionally on their fixed grid: a grid of half-width 0 -- the single tap 1 -- is the
        // identity, and its edge reset covers no pixel; velstep stays, sigma must remain finite there)
        a.freecont = 0; a.contval_fixed = 1.0;
        if (ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX) a.jax_half = 0; else a.velstep = 1e300;
    } else if (wide_stage == kWideKernels) {
        a.n_cap = ctx->wide_n_cap;                        // (arguments of the wide kernels)
    }
    return a;
}

// The finalize kernel of a tiled spectrum behind the fused kernel of `a` (adds the per-tile partials in fixed order).
int launch_finalize(mcalf_ctx* ctx, const KArgs& a, int64_t nrows, int mode, hipStream_t stream, int nparts, bool signal) {
    const int fb = 256;
    const double* partial = a.partial;
    double* out = a.out;
    long n = (long)nrows;
    int ntiles = nparts > 0 ? nparts : ctx->ntiles, m = mode, asymm = a.asymm;   // (partials per live point)
    double v4 = a.veto4, v5 = a.veto5;
    // signal: the streaming launch of a tiled spectrum -- the kernel tells the host (h_ctl[kCtlFinalized] = the launch's stamp)
    unsigned int* fin_count = signal ? &ctx->d_sctl->fin_exited : nullptr;
    unsigned int* done_word = signal ? ctx->d_ctl + kCtlFinalized : nullptr;
    unsigned int gen = ctx->stream_gen;
    void* fargs[] = {(void*)&partial, (void*)&out, (void*)&n, (void*)&ntiles, (void*)&m, (void*)&asymm, (void*)&v4, (void*)&v5,
                     (void*)&fin_count, (void*)&done_word, (void*)&gen};
    HIP_TRY(ctx, hipLaunchKernel(finalize_kernel_ptr(), dim3((unsigned)((nrows + fb - 1) / fb)), dim3(fb), fargs, 0, stream));
    return MCALF_OK;
}

static int launch_range(mcalf_ctx* ctx, int mode, const double* dP, int64_t row0, int64_t nrows, int chunk,
                        int targonly, int onecomp_fill, double* d_out, double* d_model, hipStream_t stream,
                        bool from_cube, double* d_theta, bool timed_ok, int wide_stage = kWideNone) {
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    KArgs a = make_kargs(ctx, mode, dP, row0, nrows, chunk, targonly, onecomp_fill, d_out, d_model, from_cube, d_theta, wide_stage);
    // Persistent grid = the workgroup slots of the chip (2 per CU: LDS and the 4 waves per SIMD the kernel's
    // registers allow); correctness does not depend on how many of them are resident at once.  Used once every
    // slot sees at least four items: measured on MI355X, 8 items per slot (config C) -3.5 % and 160 per slot
    // (config E) -12 % in kernel time, but 2 per slot (config B) +2 % -- there the queue and the prefetch cost
    // more than the two workgroup launches they replace, so small launches keep one workgroup per item.
    const int64_t slots = 2LL * ctx->num_cu;
    a.persist = (ctx->persist && a.nitems >= 4 * slots) ? 1 : 0;
    // Ordered hand-out while a slot sees at most 16 items: measured on MI355X, config C's spectrum, -2.2 % kernel time
    // at 8 items per slot (4096 live points) and nothing at 64 (32768), where the ordering workgroup -- its keys no
    // longer fit its registers -- would lengthen the set-up kernel by 47 us instead.
    a.order = (a.persist && ctx->ordered && ctx->selfhalo && mode != kModeOneComp && a.nitems <= 16 * slots)
                  ? ctx->d_order + row0 : nullptr;
    const dim3 grid((unsigned)(a.persist ? slots : a.nitems)), block(kBlock);
    ctx->last.persistent = a.persist; ctx->last.grid = (int32_t)grid.x; ctx->last.items = a.nitems;
    ctx->last.lines_per_sync = ctx->lps; ctx->last.selfhalo = ctx->selfhalo; ctx->last.ordered = a.order ? 1 : 0;
    // Small launches (the one-theta-at-a-time solvers, a handful of live points): ONE kernel, every workgroup sets
    // its live point up itself (mcalf_fused_kernel<..., kInline = true>) -- the set-up kernel and the dependent-launch
    // gap behind it are a fifth of such a call's latency.  Same set-up code, same bits.
    const bool inl = !a.persist && a.nitems <= ctx->inline_max_items;
    ctx->last.inline_setup = inl ? 1 : 0;
    if (!inl) {
        const int per_wg = ctx->setup_block / 64;             // live points per set-up workgroup (one wave each)
        const dim3 sgrid((unsigned)((nrows + per_wg - 1) / per_wg) + (a.order ? 1u : 0u)), sblock((unsigned)ctx->setup_block);
        long nrows_arg = (long)nrows;
        void* sargs[] = {(void*)&a, (void*)&nrows_arg};
        HIP_TRY(ctx, hipLaunchKernel(sample_kernel_ptr(ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX), sgrid, sblock, sargs, 0, stream));
    }
    const bool timed = timed_ok && ctx->profiling && ctx->ev_used + 2 <= ctx->ev.size();
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[ctx->ev_used], stream));
    {
        void* kargs[] = {(void*)&a};
        HIP_TRY(ctx, hipLaunchKernel(fused_kernel_ptr(ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX, ctx->selfhalo != 0, ctx->lps, inl),
                                     grid, block, kargs, inl ? ctx->lds_bytes_inline : ctx->lds_bytes, stream));
    }
    HIP_TRY(ctx, hipGetLastError());
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[ctx->ev_used + 1], stream));
        ctx->ev_used += 2;
    }
    if (reduces && ctx->ntiles > 1) return launch_finalize(ctx, a, nrows, mode, stream);
    return MCALF_OK;
}

// Row blocks a *_device batch is issued in.  Automatic = ONE: measured on MI355X (config C, 4096 live points), every
// extra block costs ~20 us of cross-stream event traffic and buys nothing, because the persistent fused kernel
// leaves no launch tail for the next block to fill (0.267 / 0.287 / 0.309 / 0.327 ms per batch with 1 / 2 / 3 / 4
// blocks).  The knob stays for callers that want to interleave their own work, and for the host-pointer entry,
// where blocks overlap the PCIe copies with the kernels (run_host_pipelined).
static int pick_chunks(const mcalf_ctx* ctx, int64_t batch) {
    int n = ctx->chunks_req > 0 ? ctx->chunks_req : 1;
    if (n > kMaxChunks) n = kMaxChunks;
    if ((int64_t)n > batch) n = (int)batch;
    return n < 1 ? 1 : n;
}

constexpr int kCopyStream = 1;          // aux[1]: the copy stream of the row-block pipeline; aux[0]: its second compute stream
static int ensure_aux(mcalf_ctx* ctx, int naux, bool pipeline = false) {
    if (!ctx->ev_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < naux; ++i) {
        if (!ctx->aux[i]) {
            const int rc = create_stream(ctx, &ctx->aux[i], !pipeline ? 0 : (i == kCopyStream ? +1 : -1));
            if (rc) return rc;
        }
        if (!ctx->ev_join[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
    }
    return MCALF_OK;
}

static int64_t chunk_begin(int64_t batch, int nchunks, int c) { return batch * c / nchunks; }

// Enqueue one batch on `stream` (asynchronous).  With several row blocks, blocks 1.. go to the context's
// auxiliary streams between a fork event recorded on `stream` and join events `stream` waits for, so the call
// keeps plain stream semantics for the caller (and can be captured into a hipGraph).
// Everything of a launch that can fail WITHOUT anything having been enqueued: the range check and the growth of
// the per-sample workspaces (hipMalloc).  The collective entry runs it before it enqueues anything, so that a rank
// that fails here can still take its part in the exchange (mcalf_loglike_gatherv_device).
int launch_preflight(mcalf_ctx* ctx, int mode, int64_t batch) {
    if (batch == 0) return MCALF_OK;
    if (batch < 0 || batch * (int64_t)ctx->ntiles > 0x7fff0000LL)
        return set_err(ctx, MCALF_ERR_RANGE, "batch %lld too large", (long long)batch);
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    int rc;
#ifdef MCALF_TESTING
    if (ctx->fail_preflight) return set_err(ctx, MCALF_ERR_NOMEM, "workspace growth failed (injected by MCALF_TEST_FAIL_PREFLIGHT)");
#endif
    if (reduces && ctx->ntiles > 1 && (rc = grow(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)batch * ctx->ntiles * 4)))
        return rc;
    return grow_sample_ws(ctx, batch);
}

// A batch of a WIDE-LSF context (ctx->wide: the LSF half-width does not fit a workgroup tile; numpy boundary): per pass of
// at most `rows` live points -- bounded scratch -- (1) the fused kernel WITHOUT convolution and continuum (make_kargs:
// wide_stage1) writes the unconvolved spectra into d_wide, (2) mcalf_wide_taps_kernel forms every live point's taps,
// (3) mcalf_wide_conv_kernel convolves periodically from HBM, applies the continuum and writes the model rows and / or the
// block sums of the likelihood terms, (4) mcalf_finalize_kernel adds those up.  Same entry points, same error values
// (a resolution beyond specres_max: NaN model, logL = -inf, chi2 = +inf); hires_fitter.py:445-464.
static int launch_wide(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, int targonly, int onecomp_fill, double* d_out,
                       double* d_model, hipStream_t stream, bool from_cube, double* d_theta) {
    const int64_t npix = ctx->npix, tapw = 2 * (int64_t)ctx->wide_n_cap + 1;
    const bool jax = ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX;
    const int rowlen = (mode == kModeOneComp) ? 5 : ctx->ndim;
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    const int64_t rows = wide_rows_per_pass(ctx, batch);
    if (rows < 1) return set_err(ctx, MCALF_ERR_RANGE, "wide LSF: one live point exceeds the scratch of a pass");
    const int nblocks = (int)((npix + kWideBlockPix - 1) / kWideBlockPix);
    int rc;
    if ((rc = launch_preflight(ctx, kModeModel, rows))) return rc;
    if ((rc = grow(ctx, &ctx->d_wide, &ctx->cap_wide, (size_t)rows * npix))) return rc;
    if ((rc = grow(ctx, &ctx->d_wtaps, &ctx->cap_wtaps, (size_t)rows * tapw))) return rc;
    if ((rc = grow(ctx, &ctx->d_whdr, &ctx->cap_whdr, (size_t)rows))) return rc;
    if (reduces && (rc = grow(ctx, &ctx->d_wpartial, &ctx->cap_wpartial, (size_t)rows * nblocks * 4))) return rc;
    if (mode == kModeOneComp && (rc = grow(ctx, &ctx->d_wrows, &ctx->cap_wrows, (size_t)rows * 5))) return rc;
    ctx->last.row_blocks = (int32_t)((batch + rows - 1) / rows);
    for (int64_t r0 = 0; r0 < batch; r0 += rows) {
        const int64_t n = std::min(rows, batch - r0);
        const double* P0 = dP + (size_t)r0 * rowlen;
        const double* P1 = P0;
        if (mode == kModeOneComp) {                           // (the fused kernel reads this mode's continuum from the row: rows with 1)
            double* wr = ctx->d_wrows;
            long cnt = (long)n;
            void* kargs[] = {(void*)&P0, (void*)&wr, (void*)&cnt};
            HIP_TRY(ctx, hipLaunchKernel(wide_rows_kernel_ptr(), dim3((unsigned)((n * 5 + 255) / 256)), dim3(256), kargs, 0, stream));
            P1 = wr;
        }
        if ((rc = launch_range(ctx, mode == kModeOneComp ? kModeOneComp : kModeModel, P1, 0, n, 0, targonly, onecomp_fill, nullptr, ctx->d_wide,
                               stream, from_cube, d_theta ? d_theta + (size_t)r0 * ctx->ndim : nullptr, r0 == 0, kWideFused)))
            return rc;
        KArgs a = make_kargs(ctx, mode, P0, 0, n, 0, targonly, onecomp_fill, d_out ? d_out + r0 : nullptr,
                             d_model ? d_model + (size_t)r0 * npix : nullptr, from_cube, nullptr, kWideKernels);
        a.partial = ctx->d_wpartial;
        double* taps = ctx->d_wtaps;
        const double* taps_c = taps;
        const double* flux = ctx->d_wide;
        SampleHdr* hdr = ctx->d_whdr;
        const SampleHdr* hdr_c = hdr;
        long stride = (long)tapw;
        int nb = nblocks;
        {
            void* kargs[] = {(void*)&a, (void*)&taps, (void*)&stride, (void*)&hdr};
            HIP_TRY(ctx, hipLaunchKernel(wide_taps_kernel_ptr(jax), dim3((unsigned)n), dim3(kWideBlockThreads), kargs, 0, stream));
        }
        {
            void* kargs[] = {(void*)&a, (void*)&flux, (void*)&taps_c, (void*)&stride, (void*)&hdr_c, (void*)&nb};
            HIP_TRY(ctx, hipLaunchKernel(wide_conv_kernel_ptr(jax), dim3((unsigned)nblocks, (unsigned)n), dim3(kWideBlockThreads), kargs, 0, stream));
        }
        if (reduces && (rc = launch_finalize(ctx, a, n, mode, stream, nblocks))) return rc;   // (nblocks partials per live point)
    }
    return MCALF_OK;
}

int launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, int targonly, int onecomp_fill,
           double* d_out, double* d_model, hipStream_t stream, bool from_cube, double* d_theta) {
    if (batch == 0) return MCALF_OK;
    int rc;
    if (ctx->wide) return launch_wide(ctx, mode, dP, batch, targonly, onecomp_fill, d_out, d_model, stream, from_cube, d_theta);
    // MCALF_STREAM_DEVICE=1 (diagnostic): device-pointer batches through the streaming single launch as well -- every
    // row is there from the start, so the whole grid sets up eight live points per workgroup and goes on to the items
    if (ctx->stream_device && (mode == kModeLogL || mode == kModeChi2) && !from_cube && !d_model && ctx->chunks_req <= 1 &&
        stream_qualifies(ctx, batch) && ctx->xcd_mask == (1u << kXcds) - 1u) {
        if ((rc = stream_prepare(ctx, mode, batch))) return rc;
        ctx->last.row_blocks = 1;
        if (ctx->stream_device == 2) {                     // ... with the host entries' split: a few rows eagerly, the rest by dedicated workgroups
            const int grid = 2 * ctx->num_cu, wgs = std::min((ctx->stream_wgs + kXcds - 1) / kXcds * kXcds, grid / 2 / kXcds * kXcds);
            const int64_t first_rows = ((grid - wgs) / kXcds + ctx->ntiles - 1) / ctx->ntiles;
            return stream_launch(ctx, mode, dP, batch, d_out, stream, wgs, (first_rows + 7) / 8, false, false);
        }
        return stream_launch(ctx, mode, dP, batch, d_out, stream, 0, batch, false, false);
    }
    if ((rc = launch_preflight(ctx, mode, batch))) return rc;
    const int nchunks = ctx->profiling ? 1 : pick_chunks(ctx, batch);
    ctx->last.row_blocks = nchunks;
    if (nchunks == 1)
        return launch_range(ctx, mode, dP, 0, batch, 0, targonly, onecomp_fill, d_out, d_model, stream, from_cube,
                            d_theta, true);
    if ((rc = ensure_aux(ctx, nchunks - 1))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, stream));
    for (int c = 0; c < nchunks; ++c) {
        const int64_t r0 = chunk_begin(batch, nchunks, c), r1 = chunk_begin(batch, nchunks, c + 1);
        hipStream_t st = (c == 0) ? stream : ctx->aux[c - 1];
        if (c > 0) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_fork, 0));
        if ((rc = launch_range(ctx, mode, dP, r0, r1 - r0, c, targonly, onecomp_fill, d_out, d_model, st, from_cube,
                               d_theta, false)))
            return rc;
        if (c > 0) HIP_TRY(ctx, hipEventRecord(ctx->ev_join[c - 1], st));
    }
    for (int c = 1; c < nchunks; ++c) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_join[c - 1], 0));
    return MCALF_OK;
}

extern "C" int32_t mcalf_get_chunks(const mcalf_ctx* ctx, int64_t batch) {
    if (ctx && is_multi(ctx)) return mcalf_get_chunks(ctx->subs[0], batch);
    return (ctx && batch > 0) ? pick_chunks(ctx, batch) : 0;
}

extern "C" int mcalf_set_chunks(mcalf_ctx* ctx, int32_t nchunks) {
    if (!ctx || nchunks < 0 || nchunks > kMaxChunks)
        return set_err(ctx, MCALF_ERR_INVALID, "nchunks must be 0 (automatic) .. %d", kMaxChunks);
    ctx->chunks_req = nchunks;
    for (mcalf_ctx* sub : ctx->subs) sub->chunks_req = nchunks;
    return MCALF_OK;
}

extern "C" int mcalf_profile_begin(mcalf_ctx* ctx, int32_t max_launches) {
    if (!ctx || max_launches <= 0) return set_err(ctx, MCALF_ERR_INVALID, "bad arguments");
    MCALF_SINGLE_ONLY(ctx, "mcalf_profile_begin");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    while (ctx->ev.size() < 2 * (size_t)max_launches) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev.push_back(e);
    }
    ctx->ev_used = 0;
    ctx->profiling = true;
    return MCALF_OK;
}

extern "C" int mcalf_profile_end(mcalf_ctx* ctx, double* mean_ms, int32_t* launches) {
    if (!ctx || !mean_ms) return set_err(ctx, MCALF_ERR_INVALID, "bad arguments");
    MCALF_SINGLE_ONLY(ctx, "mcalf_profile_end");
    ctx->profiling = false;
    double sum = 0.0;
    const size_t n = ctx->ev_used / 2;
    for (size_t i = 0; i < n; ++i) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2 * i + 1]));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]));
        sum += ms;
    }
    *mean_ms = n ? sum / (double)n : 0.0;
    if (launches) *launches = (int32_t)n;
    ctx->ev_used = 0;
    return MCALF_OK;
}

extern "C" int mcalf_loglike_batch_device(mcalf_ctx* ctx, const double* dP, int64_t batch, double* dlogL,
                                          void* stream) {
    if (!ctx || (batch > 0 && (!dP || !dlogL))) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    MCALF_SINGLE_ONLY(ctx, "mcalf_loglike_batch_device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->last.path = MCALF_PATH_DEVICE; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    return launch(ctx, kModeLogL, dP, batch, 0, 0, dlogL, nullptr, (hipStream_t)stream);
}

extern "C" int mcalf_model_batch_device(mcalf_ctx* ctx, const double* dP, int64_t batch, int32_t targonly,
                                        double* dflux, void* stream) {
    if (!ctx || (batch > 0 && (!dP || !dflux))) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    MCALF_SINGLE_ONLY(ctx, "mcalf_model_batch_device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->last.path = MCALF_PATH_DEVICE; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    return launch(ctx, kModeModel, dP, batch, targonly ? 1 : 0, 0, nullptr, dflux, (hipStream_t)stream);
}


// True when `p` is page-locked host memory the copy engines can read / write directly.
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();                          // ordinary pageable memory: not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// MCALF_HOST_TRACE=1 (diagnostic; accumulators in the context: host_ctx.h): where a call of the row-block pipeline spends its host
// time, printed when the context is destroyed (mean microseconds per call): staging copies by the calling thread (and the rate they
// ran at), copies its helpers took over, enqueueing split by API call, the wait for the streams, the copy of the results.
// MCALF_HOST_TRACE=2: additionally the GPU-side timeline of the LAST pipelined call -- per row block the times (us after the call
// began) at which its H2D copy started and ended and its kernels ended on the device, and when the host enqueued it.
namespace {
// The staging copy of one call, shared between the calling thread (blocks from the front, in the order the GPU wants
// them) and the helper threads (blocks from the back): whoever claims a block copies it; `copied` says it is there.
struct StageJob {
    const double* src; double* dst; int rowlen; int nchunks;
    const int64_t* bounds;
    std::atomic<int> claim[kMaxChunks];
    std::atomic<int> copied[kMaxChunks];
    std::atomic<long long> helper_bytes{0};
};
int stage_from_back(void*, int64_t, int64_t, void* arg) {
    StageJob* j = static_cast<StageJob*>(arg);
    for (int c = j->nchunks - 1; c >= 1; --c) {            // (block 0 is the calling thread's: nothing may delay it)
        int free_ = 0;
        if (!j->claim[c].compare_exchange_strong(free_, 2)) { if (free_ == 1) break; else continue; }   // met the calling thread: done
        const size_t off = (size_t)j->bounds[c] * j->rowlen, cnt = (size_t)(j->bounds[c + 1] - j->bounds[c]) * j->rowlen;
        std::memcpy(j->dst + off, j->src + off, cnt * sizeof(double));
        j->helper_bytes.fetch_add((long long)(cnt * sizeof(double)), std::memory_order_relaxed);
        j->copied[c].store(1, std::memory_order_release);
    }
    return 0;
}
}  // namespace

void host_trace_report(mcalf_ctx* ctx) {
    if (ctx->host_trace && ctx->htrace.small_calls > 0) {
        const HostTrace& t = ctx->htrace;
        const double n = (double)t.small_calls;
        std::fprintf(stderr, "mcalf host trace (zero-copy small calls, %ld calls; us per call): copy into the page-locked block %.2f, launch calls %.2f, "
                     "theta rows on the host %.2f, wait for the results %.2f, results out %.2f\n", t.small_calls,
                     t.small_prep_us / n, t.small_launch_us / n, t.small_copy_us / n, t.small_poll_us / n, t.small_out_us / n);
        if (ctx->htrace.calls == 0) ctx->htrace = HostTrace();
    }
    if (!ctx->host_trace || ctx->htrace.calls == 0) return;
    const HostTrace& t = ctx->htrace;
    const double n = (double)t.calls;
    std::fprintf(stderr, "mcalf host trace (row-block pipeline, %ld calls, %.1f blocks per call; us per call): staging copy by the caller %.1f "
                 "(%.2f GB/s), by helpers %.0f KB (waited %.1f), first block enqueued at %.1f, enqueue %.1f (of it: hipMemcpyAsync calls %.1f [longest single "
                 "call %.1f], event record + wait %.1f, launches %.1f), wait for the streams %.1f, "
                 "results out %.1f\n", t.calls, (double)t.blocks / n, t.stage_us / n, t.stage_us > 0 ? t.stage_bytes / t.stage_us * 1e-3 : 0.0,
                 t.helper_bytes / n / 1024.0, t.helper_wait_us / n, t.first_enqueued_us / n, t.enqueue_us / n, t.call_copy_us / n, t.call_copy_max,
                 t.call_order_us / n, t.call_launch_us / n, t.wait_us / n, t.out_us / n);
    ctx->htrace = HostTrace();
    const BlockTimeline& b = ctx->btrace;
    for (int c = 0; c < b.n; ++c)
        std::fprintf(stderr, "mcalf host trace, last call (%s rows), block %d: %ld rows, enqueued by the host at %.1f us (calls: copy %.1f, ordering %.1f, launches %.1f); "
                     "on the device: H2D %.1f .. %.1f, kernels done %.1f\n", b.pinned ? "page-locked" : "pageable", c, b.rows[c], b.host_enq[c], b.call_h2d[c], b.call_order[c],
                     b.call_launch[c], b.h2d0[c] * 1e3, b.h2d1[c] * 1e3, b.done[c] * 1e3);
    if (b.n) std::fprintf(stderr, "mcalf host trace, last call: the stream waits returned %.1f us after the call began\n", b.sync_us);
    ctx->btrace = BlockTimeline();
}

// The row blocks of a pipelined host-pointer call.  The FIRST block is sized by BYTES (ctx->host_first_kb KiB of
// parameter rows: the GPU starts after ~10 us of staging whatever the batch is -- as 1/8 of the batch, config E's first
// block was 0.8 MB of host memcpy before the first kernel), each following block twice the one before, the last takes
// the rest (large launches run the persistent grid and leave fewer tails).  An explicit request (mcalf_set_chunks) gives
// equal blocks, MCALF_HOST_PLAN relative sizes.
static int plan_row_blocks(const mcalf_ctx* ctx, int64_t batch, int rowlen, bool pin_in, int64_t* bounds) {
    int weights[kMaxChunks];
    int nchunks = 0;
    if (ctx->chunks_req > 0 || ctx->host_plan_n > 0) {
        if (ctx->chunks_req > 0) for (nchunks = 0; nchunks < ctx->chunks_req && nchunks < kMaxChunks; ++nchunks) weights[nchunks] = 1;
        else for (nchunks = 0; nchunks < ctx->host_plan_n; ++nchunks) weights[nchunks] = ctx->host_plan[nchunks];
        if ((int64_t)nchunks > batch) nchunks = (int)batch;
        int total = 0, run = 0;
        for (int c = 0; c < nchunks; ++c) total += weights[c];
        bounds[0] = 0;
        for (int c = 0; c < nchunks; ++c) { run += weights[c]; bounds[c + 1] = batch * run / total; }
        return nchunks;
    }
    // page-locked rows need no staging copy: the first block only has to cover the latency of its own H2D command
    int64_t first = std::max<int64_t>(64, (int64_t)ctx->host_first_kb * 1024 / ((int64_t)rowlen * (int64_t)sizeof(double)));
    if (pin_in) first *= 2;
    bounds[0] = 0;
    int64_t size = first, at = 0;
    while (nchunks < kMaxChunks - 1 && at + size + size / 2 < batch) {      // (no sliver at the end: the last block takes the rest)
        at += size;
        bounds[++nchunks] = at;
        size *= 2;
    }
    bounds[++nchunks] = batch;
    return nchunks;
}

// Large scalar-output batches through host pointers: the rows are cut into blocks; the H2D copies of ALL blocks go to a
// copy stream of their own, one behind the other as the host stages them, and block k's kernels (set-up + fused, on two
// alternating streams) wait for the event behind its copy -- so block k+1's rows are in HBM long before block k's kernels
// end, and its set-up runs in their tail.  (Round 5 queued every block's copy on the block's own stream: the copy of
// block k+2 then waited for the kernels of block k, and the GPU idled for the copy + set-up at every second block
// boundary -- measured with MCALF_HOST_TRACE=2, profiles/r06_pipeline_timeline.txt.)  Pageable caller memory
// is staged through a page-locked block of the context (the host copies block k+1 in while the GPU works on
// block k; with MCALF_STAGE_THREADS helper threads copy the batch's last blocks meanwhile); page-locked caller memory
// is used by the copy engines directly.
static int run_host_pipelined(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly,
                              int fill, double* out_scalar) {
    int rc;
    const bool trace = ctx->host_trace != 0;
    const double t_begin = trace ? now_us() : 0.0;
    const bool pin_in = is_pinned_host(P), pin_out = is_pinned_host(out_scalar);
    int64_t bounds[kMaxChunks + 1];
    int nchunks = plan_row_blocks(ctx, batch, rowlen, pin_in, bounds);
    if (ctx->profiling) { nchunks = 1; bounds[1] = batch; }
    if ((rc = ensure_aux(ctx, 2, true))) return rc;
    for (hipEvent_t& e : ctx->ev_h2d)
        if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    if (reduces && ctx->ntiles > 1 && (rc = grow(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)batch * ctx->ntiles * 4)))
        return rc;
    if ((rc = grow_sample_ws(ctx, batch))) return rc;
    const size_t need = (pin_in ? 0 : (size_t)batch * rowlen) + (pin_out ? 0 : (size_t)batch);
    if (need > ctx->cap_stage) {
        if (ctx->h_stage) HIP_TRY(ctx, hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr; ctx->cap_stage = 0;
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_stage, need * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        ctx->cap_stage = need;
    }
    double* stage_in = pin_in ? nullptr : ctx->h_stage;
    hipStream_t copy_stream = ctx->aux[kCopyStream];
    double* stage_out = pin_out ? out_scalar : ctx->h_stage + (pin_in ? 0 : (size_t)batch * rowlen);
    // Results: the kernels write logL straight into the page-locked block (its device address), 8 bytes per live
    // point over PCIe, which saves the D2H copy command of every block -- the last one is on the critical path.
    double* d_stage_out = nullptr;
    if (hipHostGetDevicePointer((void**)&d_stage_out, stage_out, 0) != hipSuccess) {
        (void)hipGetLastError();
        d_stage_out = nullptr;                            // (caller's page-locked memory that is not device-mapped)
    }
    ctx->last.path = MCALF_PATH_HOST_PIPELINED; ctx->last.row_blocks = nchunks;
    ctx->last.pinned_in = pin_in ? 1 : 0; ctx->last.pinned_out = pin_out ? 1 : 0;
    hipStream_t streams[2] = {ctx->stream, ctx->aux[0]};
    // Helpers of the staging copy (pageable rows, at least 1 MB of them): they take blocks from the back while this
    // thread stages and enqueues from the front.
    StageJob job;
    int helpers = 0;
    if (!pin_in && ctx->stage_threads > 0 && nchunks > 2 && (size_t)batch * rowlen * sizeof(double) >= ((size_t)1 << 20)) {
        job.src = P; job.dst = stage_in; job.rowlen = rowlen; job.nchunks = nchunks; job.bounds = bounds;
        for (int c = 0; c < kMaxChunks; ++c) { job.claim[c].store(0, std::memory_order_relaxed); job.copied[c].store(0, std::memory_order_relaxed); }
        while ((int)ctx->stagers.size() < ctx->stage_threads) {
            std::unique_ptr<HostWorker> w(new (std::nothrow) HostWorker());
            if (!w) break;
            w->start();
            ctx->stagers.push_back(std::move(w));
        }
        helpers = (int)ctx->stagers.size();
        for (int h = 0; h < helpers; ++h) ctx->stagers[h]->post(stage_from_back, &job, 0, 0);
    }
    // A failure in block k leaves blocks < k in flight on both streams, reading the staging block / the caller's
    // page-locked rows and writing the caller's results: never return under them (the next call may free the
    // staging block, the caller its arrays).  Every error below therefore BREAKS out of the loop and is reported behind
    // the waits for the helpers and the three streams.
    hipError_t he = hipSuccess;
    const char* what = "";
    rc = MCALF_OK;
    double t_stage = 0, t_enq = 0, t_hwait = 0, t_first = 0, stage_bytes = 0;
    const bool timeline = ctx->host_trace >= 2;
    hipEvent_t tev[3 * kMaxChunks + 1] = {};
    if (timeline) {
        for (hipEvent_t& e : tev) (void)hipEventCreate(&e);
        (void)hipEventRecord(tev[3 * kMaxChunks], ctx->stream);
        ctx->btrace = BlockTimeline();
        ctx->btrace.pinned = pin_in ? 1 : 0;
    }
    for (int c = 0; c < nchunks && rc == MCALF_OK && he == hipSuccess; ++c) {
        const int64_t r0 = bounds[c], n = bounds[c + 1] - r0;
        if (n == 0) continue;
        hipStream_t st = streams[c & 1];
        const double* src = P + (size_t)r0 * rowlen;
        const double t0 = trace ? now_us() : 0.0;
        if (!pin_in) {
            int free_ = 0;
            if (helpers == 0 || job.claim[c].compare_exchange_strong(free_, 1)) {
                std::memcpy(stage_in + (size_t)r0 * rowlen, src, (size_t)n * rowlen * sizeof(double));
                stage_bytes += (double)n * rowlen * sizeof(double);
                if (trace) t_stage += now_us() - t0;
            } else {                                      // a helper has it (or has had it): wait for its last byte
                while (job.copied[c].load(std::memory_order_acquire) == 0) __builtin_ia32_pause();
                if (trace) t_hwait += now_us() - t0;
            }
            src = stage_in + (size_t)r0 * rowlen;
        }
        const double t1 = trace ? now_us() : 0.0;
        if (timeline) (void)hipEventRecord(tev[3 * c], copy_stream);
        const double tc0 = trace ? now_us() : 0.0;
        he = hipMemcpyAsync(ctx->d_P + (size_t)r0 * rowlen, src, (size_t)n * rowlen * sizeof(double), hipMemcpyHostToDevice, copy_stream);
        if (he != hipSuccess) { what = "H2D copy of a row block"; break; }
        const double tc1 = trace ? now_us() : 0.0;
        if (timeline) (void)hipEventRecord(tev[3 * c + 1], copy_stream);
        he = hipEventRecord(ctx->ev_h2d[c], copy_stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(st, ctx->ev_h2d[c], 0);
        if (he != hipSuccess) { what = "ordering a row block's kernels behind its H2D copy"; break; }
        const double tc2 = trace ? now_us() : 0.0;
        rc = launch_range(ctx, mode, ctx->d_P, r0, n, c, targonly, fill, d_stage_out ? d_stage_out : ctx->d_out, nullptr, st,
                          false, nullptr, nchunks == 1);
        if (rc != MCALF_OK) break;
        if (!d_stage_out) {
            he = hipMemcpyAsync(stage_out + r0, ctx->d_out + r0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
            if (he != hipSuccess) { what = "D2H copy of a result block"; break; }
        }
        if (trace) {
            const double tc3 = now_us();
            ctx->htrace.call_copy_us += tc1 - tc0; ctx->htrace.call_order_us += tc2 - tc1; ctx->htrace.call_launch_us += tc3 - tc2;
            ctx->htrace.call_copy_max = std::max(ctx->htrace.call_copy_max, tc1 - tc0);
            if (timeline) { ctx->btrace.call_h2d[c] = tc1 - tc0; ctx->btrace.call_order[c] = tc2 - tc1; ctx->btrace.call_launch[c] = tc3 - tc2; }
        }
        if (timeline) { (void)hipEventRecord(tev[3 * c + 2], st); ctx->btrace.rows[c] = (long)n; ctx->btrace.host_enq[c] = now_us() - t_begin; ctx->btrace.n = c + 1; }
        if (trace) { const double t2 = now_us(); t_enq += t2 - t1; if (c == 0) t_first = t2 - t_begin; }
    }
    for (int h = 0; h < helpers; ++h) (void)ctx->stagers[h]->wait();      // (the job lives on this frame)
    const double t_w0 = trace ? now_us() : 0.0;
    const hipError_t s0 = hipStreamSynchronize(ctx->stream);
    hipError_t s1 = (nchunks > 1) ? hipStreamSynchronize(ctx->aux[0]) : hipSuccess;
    { const hipError_t s2 = hipStreamSynchronize(copy_stream); if (s1 == hipSuccess) s1 = s2; }    // (a copy no kernel waited for, after a failure)
    const double t_w1 = trace ? now_us() : 0.0;
    if (timeline) {                                       // (read and released whatever the call's outcome)
        ctx->btrace.sync_us = t_w1 - t_begin;
        for (int c = 0; c < ctx->btrace.n; ++c) {
            (void)hipEventElapsedTime(&ctx->btrace.h2d0[c], tev[3 * kMaxChunks], tev[3 * c]);
            (void)hipEventElapsedTime(&ctx->btrace.h2d1[c], tev[3 * kMaxChunks], tev[3 * c + 1]);
            (void)hipEventElapsedTime(&ctx->btrace.done[c], tev[3 * kMaxChunks], tev[3 * c + 2]);
        }
        for (hipEvent_t e : tev) if (e) (void)hipEventDestroy(e);
        (void)hipGetLastError();
    }
    if (rc != MCALF_OK) return rc;                                   // (message set by launch_range)
    if (he != hipSuccess) return set_err(ctx, MCALF_ERR_HIP, "%s failed: %s", what, hipGetErrorString(he));
    if (s0 != hipSuccess || s1 != hipSuccess)
        return set_err(ctx, MCALF_ERR_HIP, "stream synchronisation failed: %s", hipGetErrorString(s0 != hipSuccess ? s0 : s1));
    if (!pin_out) std::memcpy(out_scalar, stage_out, (size_t)batch * sizeof(double));
    if (trace) {
        HostTrace& t = ctx->htrace;
        t.calls++; t.blocks += nchunks; t.stage_us += t_stage; t.stage_bytes += stage_bytes; t.helper_wait_us += t_hwait;
        t.helper_bytes += helpers ? (double)job.helper_bytes.load() : 0.0; t.enqueue_us += t_enq; t.first_enqueued_us += t_first;
        t.wait_us += t_w1 - t_w0; t.out_us += now_us() - t_w1;
    }
    return MCALF_OK;
}

// The page-locked, device-mapped block small calls go through: parameters in its first half, results in its second.
int ensure_small(mcalf_ctx* ctx) {
    if (!ctx->h_small) {
        // (coherent: the host fills the result slots before a launch and reads them while it runs)
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_small, 2 * kSmallDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->d_small, ctx->h_small, 0));
    }
    return MCALF_OK;
}

// Small scalar-output calls (up to kSmallDoubles parameters: single-theta calls, config B's batch), zero-copy: a
// single-theta call is dominated by the latency of its two copy commands.  from_cube / theta_out: as in run_host_stream.
static int run_host_small(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly, int fill,
                          double* out_scalar, bool from_cube, double* theta_out) {
    int rc;
    if (resident_serves(ctx, mode, batch, rowlen, from_cube)) return resident_call(ctx, P, rowlen, out_scalar);
    if ((rc = ensure_small(ctx))) return rc;
    const bool trace = ctx->host_trace != 0;
    const double ts0 = trace ? now_us() : 0.0;
    // (Copy first, launch afterwards.  Round 6 tried the other order for calls of >= 32 KB -- the set-up kernel's workgroups
    // waiting for the host's row count, so that the launch latency would run under the copy: the copy of config B's 200 KB
    // takes 1.5 us, the waits' PCIe looks cost 8 us; docs/optimisation_log.md 8f.)
    std::memcpy(ctx->h_small, P, (size_t)batch * rowlen * sizeof(double));
    ctx->last.path = MCALF_PATH_HOST_ZEROCOPY; ctx->last.pinned_in = ctx->last.pinned_out = 0; ctx->last.stream_fallback = 0;
    // Completion is read off the results: their slots are filled with a NaN no kernel produces, and the call is over
    // when none is left -- a stream wait costs an interrupt and a thread wake-up on top of the kernel, a fifth of a
    // one-theta call.  (The stream is asked now and then, so that a failed launch cannot keep the call here.)
    uint64_t* res = reinterpret_cast<uint64_t*>(ctx->h_small + kSmallDoubles);
    const bool poll = ctx->stream_poll != 0;
    if (poll)
        for (int64_t i = 0; i < batch; ++i) __atomic_store_n(res + i, kResultPending, __ATOMIC_RELEASE);
    const double ts1 = trace ? now_us() : 0.0;
    rc = launch(ctx, mode, ctx->d_small, batch, targonly, fill, ctx->d_small + kSmallDoubles, nullptr, ctx->stream, from_cube);
    if (rc) return rc;
    const double ts2 = trace ? now_us() : 0.0;
    if (theta_out) host_scale_cube(ctx, P, batch, theta_out);       // (under the launch)
    const double ts3 = trace ? now_us() : 0.0;
    bool done = false;
    if (poll) {
        int64_t left = batch;                            // results [left, batch) have been seen
        for (unsigned long spins = 1;; ++spins) {
            while (left > 0 && __atomic_load_n(res + left - 1, __ATOMIC_ACQUIRE) != kResultPending) --left;
            if (left == 0) { done = true; break; }
            if ((spins & 0x3FFFul) == 0 && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;
            __builtin_ia32_pause();
        }
    }
    if (!done) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->last.stream_polled = done ? 1 : 0;
    const double ts4 = trace ? now_us() : 0.0;
    std::memcpy(out_scalar, ctx->h_small + kSmallDoubles, (size_t)batch * sizeof(double));
    if (trace) {
        HostTrace& t = ctx->htrace;
        t.small_calls++; t.small_prep_us += ts1 - ts0; t.small_launch_us += ts2 - ts1; t.small_copy_us += ts3 - ts2;
        t.small_poll_us += ts4 - ts3; t.small_out_us += now_us() - ts4;
    }
    return MCALF_OK;
}

// Host-pointer entries: stage through the context's workspaces on its private stream.
static int run_host(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly, int fill,
                    double* out_scalar, double* out_model);
namespace {
struct HostShard { int mode; const double* P; int rowlen, targonly, fill; double *out_scalar, *out_model; int64_t npix; };
int host_shard(void* sub, int64_t lo, int64_t hi, void* arg) {
    const HostShard* c = static_cast<const HostShard*>(arg);
    return run_host(static_cast<mcalf_ctx*>(sub), c->mode, c->P + (size_t)lo * c->rowlen, hi - lo, c->rowlen, c->targonly, c->fill,
                    c->out_scalar ? c->out_scalar + lo : nullptr, c->out_model ? c->out_model + (size_t)lo * c->npix : nullptr);
}
}  // namespace
static int run_host(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly, int fill,
                    double* out_scalar, double* out_model) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (batch < 0) return set_err(ctx, MCALF_ERR_INVALID, "negative batch");
    if (batch == 0) return MCALF_OK;
    if (!P || (!out_scalar && !out_model)) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    if (is_multi(ctx)) {                                  // contiguous row blocks, one per device, straight into the caller's arrays
        HostShard c = {mode, P, rowlen, targonly, fill, out_scalar, out_model, (int64_t)ctx->npix};
        return multi_run(ctx, batch, host_shard, &c);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if (out_scalar && !out_model && (size_t)batch * rowlen <= kSmallDoubles && (size_t)batch <= kSmallDoubles) {
        // (a batch that is small in bytes but reaches the streaming launch's item count -- MCALF_STREAM_MIN items per
        // workgroup slot -- streams: one launch that sets its live points up itself, instead of set-up + fused kernel)
        if ((mode == kModeLogL || mode == kModeChi2) && stream_qualifies(ctx, batch)) {
            bool taken = false;
            if ((rc = run_host_stream(ctx, mode, P, batch, rowlen, out_scalar, &taken)) != MCALF_OK || taken) return rc;
        }
        return run_host_small(ctx, mode, P, batch, rowlen, targonly, fill, out_scalar, false, nullptr);
    }
    if ((rc = grow(ctx, &ctx->d_P, &ctx->cap_P, (size_t)batch * rowlen))) return rc;
    if (out_scalar && (rc = grow(ctx, &ctx->d_out, &ctx->cap_out, (size_t)batch))) return rc;
    if (out_model && (rc = grow(ctx, &ctx->d_model, &ctx->cap_model, (size_t)batch * ctx->npix))) return rc;
    if (out_scalar && !out_model && !ctx->wide) {
        if (mode == kModeLogL || mode == kModeChi2) {
            bool taken = false;
            if ((rc = run_host_stream(ctx, mode, P, batch, rowlen, out_scalar, &taken)) != MCALF_OK || taken) return rc;
        }
        return run_host_pipelined(ctx, mode, P, batch, rowlen, targonly, fill, out_scalar);
    }
    ctx->last.path = MCALF_PATH_HOST_STAGED; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_P, P, (size_t)batch * rowlen * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream));
    rc = launch(ctx, mode, ctx->d_P, batch, targonly, fill, out_scalar ? ctx->d_out : nullptr,
                out_model ? ctx->d_model : nullptr, ctx->stream);
    if (rc) return rc;
    if (out_scalar)
        HIP_TRY(ctx, hipMemcpyAsync(out_scalar, ctx->d_out, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost,
                                    ctx->stream));
    if (out_model)
        HIP_TRY(ctx, hipMemcpyAsync(out_model, ctx->d_model, (size_t)batch * ctx->npix * sizeof(double),
                                    hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCALF_OK;
}

extern "C" int mcalf_loglike_batch(mcalf_ctx* ctx, const double* P, int64_t batch, double* logL) {
    return run_host(ctx, kModeLogL, P, batch, ctx ? ctx->ndim : 0, 0, 0, logL, nullptr);
}

extern "C" int mcalf_chi2_batch(mcalf_ctx* ctx, const double* P, int64_t batch, double* chi2) {
    return run_host(ctx, kModeChi2, P, batch, ctx ? ctx->ndim : 0, 0, 0, chi2, nullptr);
}

extern "C" int mcalf_model_batch(mcalf_ctx* ctx, const double* P, int64_t batch, int32_t targonly, double* flux) {
    return run_host(ctx, kModeModel, P, batch, ctx ? ctx->ndim : 0, targonly ? 1 : 0, 0, nullptr, flux);
}

extern "C" int mcalf_onecomp_batch(mcalf_ctx* ctx, const double* Q, int64_t batch, int32_t which, double* flux) {
    if (ctx && (which < 0 || which >= 2 + ctx->nlines))
        return set_err(ctx, MCALF_ERR_INVALID, "onecomp: `which` must be 0 (all lines), 1 (filler) or 2+k with k < %d",
                       ctx->nlines);
    return run_host(ctx, kModeOneComp, Q, batch, 5, 0, which, nullptr, flux);
}

extern "C" int mcalf_set_prior(mcalf_ctx* ctx, const double* lo, const double* hi, int32_t int_ncomp) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (!lo || !hi) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    if (is_multi(ctx)) {
        for (mcalf_ctx* sub : ctx->subs) {
            const int rc = mcalf_set_prior(sub, lo, hi, int_ncomp);
            if (rc) return set_err(ctx, rc, "%s", sub->err.c_str());
        }
        ctx->prior_set = true;
        return MCALF_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_prior) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_prior, 2 * (size_t)ctx->ndim * sizeof(double)));
    // synchronous copies: the caller's arrays are borrowed for this call only, and a later *_device call may
    // run on any stream
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_prior, lo, ctx->ndim * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_prior + ctx->ndim, hi, ctx->ndim * sizeof(double), hipMemcpyHostToDevice));
    ctx->h_prior.assign(lo, lo + ctx->ndim);
    ctx->h_prior.insert(ctx->h_prior.end(), hi, hi + ctx->ndim);
    ctx->prior_set = true;
    ctx->prior_int = int_ncomp ? 1 : 0;
    return MCALF_OK;
}

extern "C" int mcalf_loglike_cube_batch_device(mcalf_ctx* ctx, const double* dcube, int64_t batch, double* dtheta,
                                               double* dlogL, void* stream) {
    if (!ctx || (batch > 0 && (!dcube || !dlogL))) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    MCALF_SINGLE_ONLY(ctx, "mcalf_loglike_cube_batch_device");
    if (!ctx->prior_set) return set_err(ctx, MCALF_ERR_INVALID, "mcalf_set_prior has not been called");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->last.path = MCALF_PATH_DEVICE; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    return launch(ctx, kModeLogL, dcube, batch, 0, 0, dlogL, nullptr, (hipStream_t)stream, true, dtheta);
}

extern "C" int mcalf_loglike_cube_batch(mcalf_ctx* ctx, const double* cube, int64_t batch, double* theta,
                                        double* logL) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (batch < 0) return set_err(ctx, MCALF_ERR_INVALID, "negative batch");
    if (!ctx->prior_set) return set_err(ctx, MCALF_ERR_INVALID, "mcalf_set_prior has not been called");
    if (batch == 0) return MCALF_OK;
    if (!cube || !logL) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    if (is_multi(ctx)) {
        struct CubeShard { const double* cube; double *theta, *logL; int ndim; } c = {cube, theta, logL, ctx->ndim};
        return multi_run(ctx, batch, [](void* sub, int64_t lo, int64_t hi, void* arg) {
            const CubeShard* q = static_cast<const CubeShard*>(arg);
            return mcalf_loglike_cube_batch(static_cast<mcalf_ctx*>(sub), q->cube + (size_t)lo * q->ndim, hi - lo, q->theta ? q->theta + (size_t)lo * q->ndim : nullptr, q->logL + lo);
        }, &c);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t total = (size_t)batch * ctx->ndim;
    int rc;
    // the paths of mcalf_loglike_batch, with the prior transform applied while the rows are decoded and the transformed
    // rows formed on the host under the launch: the zero-copy small call, ONE streaming launch for large batches
    if (total <= kSmallDoubles && (size_t)batch <= kSmallDoubles) {
        if (stream_qualifies(ctx, batch)) {
            bool taken = false;
            if ((rc = run_host_stream(ctx, kModeLogL, cube, batch, ctx->ndim, logL, &taken, true, theta)) != MCALF_OK || taken) return rc;
        }
        return run_host_small(ctx, kModeLogL, cube, batch, ctx->ndim, 0, 0, logL, true, theta);
    }
    {
        bool taken = false;
        if ((rc = run_host_stream(ctx, kModeLogL, cube, batch, ctx->ndim, logL, &taken, true, theta)) != MCALF_OK || taken) return rc;
    }
    // otherwise (tiled spectra, explicit row blocks): staged copies, the transformed rows come back from the device
    ctx->last.path = MCALF_PATH_HOST_STAGED; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    if ((rc = grow(ctx, &ctx->d_P, &ctx->cap_P, total))) return rc;
    if ((rc = grow(ctx, &ctx->d_out, &ctx->cap_out, (size_t)batch))) return rc;
    if (theta && (rc = grow(ctx, &ctx->d_model, &ctx->cap_model, total))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_P, cube, total * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = launch(ctx, kModeLogL, ctx->d_P, batch, 0, 0, ctx->d_out, nullptr, ctx->stream, true,
                theta ? ctx->d_model : nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(logL, ctx->d_out, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (theta)
        HIP_TRY(ctx, hipMemcpyAsync(theta, ctx->d_model, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCALF_OK;
}

extern "C" int mcalf_scale_cube_batch(mcalf_ctx* ctx, const double* lo, const double* hi, const double* cube,
                                      int64_t batch, int32_t int_ncomp, double* theta) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (batch < 0) return set_err(ctx, MCALF_ERR_INVALID, "negative batch");
    if (batch == 0) return MCALF_OK;
    if (!lo || !hi || !cube || !theta) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    if (is_multi(ctx)) return mcalf_scale_cube_batch(ctx->subs[0], lo, hi, cube, batch, int_ncomp, theta);   // (a few bytes per row: one device)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t total = (size_t)batch * ctx->ndim;
    int rc;
    if ((rc = grow(ctx, &ctx->d_P, &ctx->cap_P, total))) return rc;
    if ((rc = grow(ctx, &ctx->d_model, &ctx->cap_model, total))) return rc;
    if (!ctx->d_bounds) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_bounds, 2 * (size_t)ctx->ndim * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bounds, lo, ctx->ndim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bounds + ctx->ndim, hi, ctx->ndim * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_P, cube, total * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    {
        const double *lo_d = ctx->d_bounds, *hi_d = ctx->d_bounds + ctx->ndim, *cube_d = ctx->d_P;
        long n = (long)total;
        int nd = ctx->ndim, slot = ctx->startind, as_int = int_ncomp;
        double* theta_d = ctx->d_model;
        void* kargs[] = {(void*)&lo_d, (void*)&hi_d, (void*)&cube_d, (void*)&n, (void*)&nd, (void*)&slot, (void*)&as_int, (void*)&theta_d};
        HIP_TRY(ctx, hipLaunchKernel(scale_cube_kernel_ptr(), dim3((unsigned)((total + 255) / 256)), dim3(256), kargs, 0, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(theta, ctx->d_model, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCALF_OK;
}

static int hjerting_impl(const double* x, const double* y, int64_t n, double* out, int32_t device, int node_form) {
    if (n < 0 || (n > 0 && (!x || !y || !out))) return set_err(nullptr, MCALF_ERR_INVALID, "bad arguments");
    if (n == 0) return MCALF_OK;
    int dev = 0;
    int rc = pick_device(nullptr, device, &dev, nullptr);
    if (rc) return rc;
    HIP_TRY(nullptr, hipSetDevice(dev));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr, *dtabs = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    hipError_t em = hipMalloc((void**)&dx, nb);
    if (em == hipSuccess) em = hipMalloc((void**)&dy, nb);
    if (em == hipSuccess) em = hipMalloc((void**)&dout, nb);
    if (em != hipSuccess) rc = set_err(nullptr, MCALF_ERR_HIP, "hjerting: hipMalloc failed: %s", hipGetErrorString(em));
    else rc = upload_tables(nullptr, &dtabs);
    if (rc == MCALF_OK) {
        hipError_t e = hipMemcpy(dx, x, nb, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dy, y, nb, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            long cnt = (long)n;
            void* kargs[] = {(void*)&dx, (void*)&dy, (void*)&cnt, (void*)&dout, (void*)&dtabs, (void*)&node_form};
            e = hipLaunchKernel(hjert_kernel_ptr(), dim3((unsigned)((n + 255) / 256)), dim3(256), kargs, 0, nullptr);
        }
        if (e == hipSuccess) e = hipMemcpy(out, dout, nb, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_err(nullptr, MCALF_ERR_HIP, "hjerting: %s", hipGetErrorString(e));
    }
    for (double* b : {dx, dy, dout, dtabs})
        if (b) (void)hipFree(b);
    return rc;
}

extern "C" int mcalf_voigt_hjerting(const double* x, const double* y, int64_t n, double* out, int32_t device) {
    return hjerting_impl(x, y, n, out, device, 0);
}

extern "C" int mcalf_voigt_hjerting_nodes(const double* x, const double* y, int64_t n, double* out, int32_t device) {
    return hjerting_impl(x, y, n, out, device, 1);
}


