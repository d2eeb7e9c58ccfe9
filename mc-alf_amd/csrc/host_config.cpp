// libmcalf_hip.so, host side: the library's configuration.  The ONLY place that reads the environment: mcalf_create takes
// one snapshot per context (read_environment), applies it to the context's built-in values (apply_environment) and
// mcalf_get_config prints the values the context actually runs under -- a library whose default path depends on the
// environment of whoever loads it must at least be able to say what it found there.
#include "host_ctx.h"

namespace {
bool env_int(const char* name, int* out) {
    const char* e = std::getenv(name);
    if (!e || !*e) return false;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 0);
    if (end == e) return false;
    *out = (int)std::min<long>(std::max<long>(v, -2147483647L), 2147483647L);
    return true;
}
}  // namespace

EnvKnobs read_environment() {
    EnvKnobs k;
    int v;
    if (env_int("MCALF_LINES_PER_SYNC", &v) && (v == 4 || v == 5)) k.lines_per_sync = v;
    if (env_int("MCALF_PERSIST", &v)) k.persist = v != 0;
    if (env_int("MCALF_ORDER", &v)) k.order = v != 0;
    if (env_int("MCALF_INLINE_MAX", &v)) k.inline_max = std::max(0, v);
    if (env_int("MCALF_RESIDENT_US", &v)) k.resident_us = std::min(1000000, std::max(0, v));
    if (env_int("MCALF_SETUP_BLOCK", &v) && v >= 64 && v <= kSetupBlockMax && v % 64 == 0) k.setup_block = v;
    if (const char* hp = std::getenv("MCALF_HOST_PLAN")) {            // e.g. "1,3,4": relative block sizes (diagnostic)
        for (const char* q = hp; *q && k.host_plan_n < kMaxChunks;) {
            const int w = std::atoi(q);
            if (w > 0) k.host_plan[k.host_plan_n++] = w;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    if (env_int("MCALF_HOST_FIRST_KB", &v) && v > 0) k.host_first_kb = std::min(v, 1 << 20);
    if (env_int("MCALF_HOST_TRACE", &v)) k.host_trace = std::min(std::max(v, 0), 2);
    if (env_int("MCALF_STAGE_THREADS", &v)) k.stage_threads = std::min(std::max(v, 0), 8);
    if (env_int("MCALF_STREAM", &v)) k.stream = std::min(std::max(v, 0), 2);
    if (env_int("MCALF_STREAM_WGS", &v)) k.stream_wgs = std::max(v, 1);
    if (env_int("MCALF_STREAM_MIN", &v)) k.stream_min = std::min(std::max(v, 1), 64);
    if (env_int("MCALF_STREAM_POLL", &v)) k.stream_poll = v != 0;
    if (env_int("MCALF_STREAM_EAGER", &v)) k.stream_eager = std::max(v, 0);
    if (env_int("MCALF_STREAM_CHUNK", &v)) k.stream_chunk = std::min(std::max(v & ~7, 8), 512);
    if (env_int("MCALF_STREAM_DEVICE", &v)) k.stream_device = std::min(std::max(v, 0), 2);
    if (env_int("MCALF_STREAM_TRACE", &v)) k.stream_trace = v != 0;
    if (const char* e = std::getenv("MCALF_STREAM_TIMEOUT")) {
        const double t = std::atof(e);
        if (t > 0.0 && t <= 60.0) k.stream_timeout_s = t;
    }
    if (env_int("MCALF_CHUNKS", &v) && v >= 0 && v <= kMaxChunks) k.chunks = v;
    if (const char* e = std::getenv("MCALF_RCCL_LIB")) k.rccl_lib = e;
#ifdef MCALF_TESTING
    if (env_int("MCALF_TEST_FAIL_PREFLIGHT", &v)) k.test_fail_preflight = v != 0;
    if (const char* t = std::getenv("MCALF_TEST_XCD_MASK")) k.test_xcd_mask = (long)std::strtoul(t, nullptr, 0);
    if (env_int("MCALF_TEST_STARVE", &v)) k.test_starve = v != 0;
#endif
    return k;
}

// The environment's values over the context's built-in ones (everything here is independent of the device; what depends
// on it -- inline_max_items, stream_wgs against the CU count -- is finished by create_impl once the device is known).
void apply_environment(mcalf_ctx* ctx) {
    const EnvKnobs& k = ctx->env;
    if (k.persist >= 0) ctx->persist = k.persist;
    if (k.order >= 0) ctx->ordered = k.order;
    if (k.resident_us >= 0) ctx->resident_us = k.resident_us;
    if (k.setup_block >= 0) ctx->setup_block = k.setup_block;
    for (int i = 0; i < k.host_plan_n; ++i) ctx->host_plan[i] = k.host_plan[i];
    ctx->host_plan_n = k.host_plan_n;
    if (k.host_first_kb >= 0) ctx->host_first_kb = k.host_first_kb;
    if (k.host_trace >= 0) ctx->host_trace = k.host_trace;
    if (k.stage_threads >= 0) ctx->stage_threads = k.stage_threads;
    if (k.stream >= 0) ctx->stream_on = k.stream;
    if (k.stream_min >= 0) ctx->stream_min = k.stream_min;
    if (k.stream_poll >= 0) ctx->stream_poll = k.stream_poll;
    if (k.stream_eager >= 0) ctx->stream_eager = k.stream_eager;
    if (k.stream_chunk >= 0) ctx->stream_chunk = k.stream_chunk;
    if (k.stream_device >= 0) ctx->stream_device = k.stream_device;
    if (k.stream_trace >= 0) ctx->stream_trace = k.stream_trace;
    if (k.stream_timeout_s > 0.0) ctx->stream_timeout_s = k.stream_timeout_s;
    if (k.chunks >= 0) ctx->chunks_req = k.chunks;
#ifdef MCALF_TESTING
    if (k.test_fail_preflight >= 0) ctx->fail_preflight = k.test_fail_preflight != 0;
#endif
}

// "name=value" pairs, space separated, of everything a knob decides for this context; `[env: ...]` lists the variables
// that were set (and valid) when the context was created.
extern "C" int mcalf_get_config(const mcalf_ctx* ctx, char* buf, int64_t n) {
    if (!ctx || !buf || n <= 0) return set_err(nullptr, MCALF_ERR_INVALID, "mcalf_get_config: NULL argument or empty buffer");
    const EnvKnobs& k = ctx->env;
    std::string s;
    char tmp[256];
    auto add = [&](const char* fmt, auto... args) { snprintf(tmp, sizeof tmp, fmt, args...); s += tmp; };
    add("devices=%d lines_per_sync=%d persist=%d order=%d inline_max=%d resident_us=%d setup_block=%d chunks=%d ", (int)ctx->subs.size() > 0 ? (int)ctx->subs.size() : 1,
        ctx->lps, ctx->persist, ctx->ordered, ctx->inline_max_items, ctx->resident_us, ctx->setup_block, ctx->chunks_req);
    s += "host_plan=";
    if (ctx->host_plan_n == 0) s += "auto";
    for (int i = 0; i < ctx->host_plan_n; ++i) add(i ? ",%d" : "%d", ctx->host_plan[i]);
    add(" host_first_kb=%d host_trace=%d stage_threads=%d ", ctx->host_first_kb, ctx->host_trace, ctx->stage_threads);
    add("stream=%d stream_min=%d stream_wgs=%d stream_poll=%d stream_eager=%d stream_chunk=%d stream_device=%d stream_trace=%d stream_timeout_s=%g ",
        ctx->stream_on, ctx->stream_min, ctx->stream_wgs, ctx->stream_poll, ctx->stream_eager, ctx->stream_chunk, ctx->stream_device, ctx->stream_trace,
        ctx->stream_timeout_s);
    add("xcd_mask=0x%x cu_mask_words=%d wide_lsf=%d", ctx->xcd_mask, (int)ctx->cu_mask.size(), ctx->wide);
    s += " [env:";
    auto named = [&](bool set, const char* name) { if (set) { s += ' '; s += name; } };
    named(k.lines_per_sync >= 0, "MCALF_LINES_PER_SYNC"); named(k.persist >= 0, "MCALF_PERSIST"); named(k.order >= 0, "MCALF_ORDER");
    named(k.inline_max >= 0, "MCALF_INLINE_MAX"); named(k.resident_us >= 0, "MCALF_RESIDENT_US"); named(k.setup_block >= 0, "MCALF_SETUP_BLOCK");
    named(k.host_plan_n > 0, "MCALF_HOST_PLAN"); named(k.host_first_kb >= 0, "MCALF_HOST_FIRST_KB"); named(k.host_trace >= 0, "MCALF_HOST_TRACE");
    named(k.stage_threads >= 0, "MCALF_STAGE_THREADS"); named(k.stream >= 0, "MCALF_STREAM"); named(k.stream_min >= 0, "MCALF_STREAM_MIN"); named(k.stream_wgs >= 0, "MCALF_STREAM_WGS");
    named(k.stream_poll >= 0, "MCALF_STREAM_POLL"); named(k.stream_eager >= 0, "MCALF_STREAM_EAGER"); named(k.stream_chunk >= 0, "MCALF_STREAM_CHUNK");
    named(k.stream_device >= 0, "MCALF_STREAM_DEVICE"); named(k.stream_trace >= 0, "MCALF_STREAM_TRACE");
    named(k.stream_timeout_s > 0.0, "MCALF_STREAM_TIMEOUT"); named(k.chunks >= 0, "MCALF_CHUNKS"); named(!k.rccl_lib.empty(), "MCALF_RCCL_LIB");
    if (s.back() == ':') s += " none";
    s += "]";
    snprintf(buf, (size_t)n, "%s", s.c_str());
    return MCALF_OK;
}
