// libmcalf_hip.so, host side: ONE process driving several devices (mcalf_create_multi).
//
// The reference's large batches arise inside one Python process (jaxns vmaps the likelihood over the live points:
// cli.py:274-280), and its only parallelism is data parallelism over live points (cli.py:110).  A multi-device context
// is that for the host-pointer entries: one complete sub-context per device entry (spectrum replicated), contiguous
// row blocks of the caller's batch (shard_bounds of mc-alf_amd/dist.py: block sizes differ by at most one, the first
// batch % n sub-contexts take the longer ones), every device's call -- streaming launch, row-block pipeline, zero-copy small call, whatever its
// shard's size selects -- issued concurrently, each writing its block of results straight into the caller's array.
// No collective: the results land in host memory, where the caller wants them.  Because a live point's arithmetic does
// not depend on the shard, the batch equals the single-device result bit for bit.
//
// The host-pointer entries of a sub-context are synchronous and busy (staging copy, polling), so concurrency needs a
// thread per device: the calling thread drives sub-context 0, a persistent worker thread each of the others (spins for
// a short while after a job -- a sampler calls back to back -- then sleeps on a condition variable).
#include <new>

#include "host_ctx.h"

namespace {
constexpr int kMaxDevices = 16;             // (mcalf_info_t.devices)
constexpr int64_t kMultiMinRows = 256;      // a device is given at least this many rows: below, a shard's call is all latency
}  // namespace

struct MultiPool {
    std::vector<std::unique_ptr<HostWorker>> workers;        // worker k - 1 drives sub-context k
};

void multi_release(mcalf_ctx* ctx) {
    if (ctx->pool) {
        for (auto& w : ctx->pool->workers) w->stop();
        delete ctx->pool;
        ctx->pool = nullptr;
    }
    for (mcalf_ctx* s : ctx->subs) mcalf_destroy(s);
    ctx->subs.clear();
}

int multi_active(const mcalf_ctx* ctx, int64_t batch) {
    const int64_t n = std::min<int64_t>((int64_t)ctx->subs.size(), batch / kMultiMinRows);
    return (int)std::max<int64_t>(1, n);                  // (mcalf_shard_bounds states the same rule for callers and tests)
}

int multi_run(mcalf_ctx* ctx, int64_t batch, WorkFn fn, void* arg) {
    const int n = multi_active(ctx, batch);
    ctx->multi_last_active = n;
    int64_t lo, hi;
    for (int k = 1; k < n; ++k) {
        multi_bounds(batch, n, k, &lo, &hi);
        ctx->pool->workers[k - 1]->post(fn, arg, lo, hi);
    }
    multi_bounds(batch, n, 0, &lo, &hi);
    int rc = fn(ctx->subs[0], lo, hi, arg);
    int bad = rc != MCALF_OK ? 0 : -1;
    for (int k = 1; k < n; ++k) {                             // (every shard is waited for, whatever the others returned:
        const int rk = ctx->pool->workers[k - 1]->wait();     // they write into the caller's arrays)
        if (rk != MCALF_OK && bad < 0) { rc = rk; bad = k; }
    }
    if (bad >= 0) return set_err(ctx, rc, "device entry %d (HIP device %d): %s", bad, ctx->subs[bad]->device, ctx->subs[bad]->err.c_str());
    return MCALF_OK;
}

extern "C" int mcalf_create_multi(const mcalf_spec* spec, const int32_t* devices, int32_t ndevices, mcalf_ctx** out) {
    if (!out) return set_err(nullptr, MCALF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!spec || !devices || ndevices < 1 || ndevices > kMaxDevices)
        return set_err(nullptr, MCALF_ERR_INVALID, "mcalf_create_multi: spec / devices non-NULL, 1 <= ndevices <= %d", kMaxDevices);
    mcalf_ctx* ctx = new (std::nothrow) mcalf_ctx();
    if (!ctx) return set_err(nullptr, MCALF_ERR_NOMEM, "out of host memory");
    for (int k = 0; k < ndevices; ++k) {
        mcalf_spec sp = *spec;
        sp.device = devices[k];
        mcalf_ctx* sub = nullptr;
        const int rc = mcalf_create(&sp, &sub);               // (leaves the message in the thread's slot)
        if (rc != MCALF_OK) {
            const std::string why = mcalf_last_error(nullptr);
            multi_release(ctx);
            delete ctx;
            return set_err(nullptr, rc, "mcalf_create_multi, device entry %d (HIP device %d): %s", k, devices[k], why.c_str());
        }
        ctx->subs.push_back(sub);
    }
    // the parent owns no device memory: its problem / geometry / knob fields mirror sub-context 0 (mcalf_info, mcalf_get_config)
    const mcalf_ctx* s0 = ctx->subs[0];
    ctx->env = s0->env;
    ctx->device = s0->device; ctx->arch = s0->arch;
    ctx->npix = s0->npix; ctx->nlines = s0->nlines; ctx->ncompmax = s0->ncompmax; ctx->nfill = s0->nfill;
    ctx->freespecres = s0->freespecres; ctx->freecont = s0->freecont; ctx->conv_mode = s0->conv_mode;
    ctx->ndim = s0->ndim; ctx->startind = s0->startind; ctx->endind = s0->endind;
    ctx->specres_fixed = s0->specres_fixed; ctx->specres_max = s0->specres_max; ctx->contval_fixed = s0->contval_fixed;
    ctx->velstep = s0->velstep; ctx->asymm = s0->asymm; ctx->veto4 = s0->veto4; ctx->veto5 = s0->veto5;
    ctx->n_cap = s0->n_cap; ctx->tile = s0->tile; ctx->ntiles = s0->ntiles; ctx->ncl_cap = s0->ncl_cap; ctx->jax_half = s0->jax_half;
    ctx->selfhalo = s0->selfhalo; ctx->lps = s0->lps; ctx->wide = s0->wide; ctx->wide_n_cap = s0->wide_n_cap;
    ctx->num_cu = s0->num_cu; ctx->xcd_mask = s0->xcd_mask;
    ctx->pool = new (std::nothrow) MultiPool();
    if (!ctx->pool) { multi_release(ctx); delete ctx; return set_err(nullptr, MCALF_ERR_NOMEM, "out of host memory"); }
    for (int k = 1; k < ndevices; ++k) {
        std::unique_ptr<HostWorker> w(new HostWorker());
        w->who = ctx->subs[k];
        w->device = ctx->subs[k]->device;
        w->start();
        ctx->pool->workers.push_back(std::move(w));
    }
    *out = ctx;
    return MCALF_OK;
}

extern "C" int mcalf_shard_bounds(int64_t batch, int32_t nentries, int32_t k, int32_t* entries_used, int64_t* lo, int64_t* hi) {
    if (batch < 0 || nentries < 1 || nentries > kMaxDevices || !entries_used || !lo || !hi)
        return set_err(nullptr, MCALF_ERR_INVALID, "mcalf_shard_bounds: batch >= 0, 1 <= nentries <= %d, non-NULL outputs", kMaxDevices);
    const int n = (int)std::max<int64_t>(1, std::min<int64_t>(nentries, batch / kMultiMinRows));      // (multi_active)
    *entries_used = n;
    if (k < 0 || k >= nentries) return set_err(nullptr, MCALF_ERR_INVALID, "mcalf_shard_bounds: entry %d of %d", k, nentries);
    if (k >= n) { *lo = *hi = batch; return MCALF_OK; }                                                // (an entry that sits this call out)
    multi_bounds(batch, n, k, lo, hi);
    return MCALF_OK;
}

extern "C" int mcalf_last_launch_sub(const mcalf_ctx* ctx, int32_t k, mcalf_launch_info_t* info) {
    if (!ctx || !info) return set_err(nullptr, MCALF_ERR_INVALID, "NULL argument");
    if (!is_multi(ctx)) return k == 0 ? mcalf_last_launch(ctx, info) : set_err(nullptr, MCALF_ERR_INVALID, "a single-device context has sub-context 0 only");
    if (k < 0 || k >= (int32_t)ctx->subs.size()) return set_err(nullptr, MCALF_ERR_INVALID, "sub-context %d of %zu", k, ctx->subs.size());
    return mcalf_last_launch(ctx->subs[k], info);
}
