// Host side of libmcalf_hip.so, shared by its translation units (host_abi.cpp, host_stream.cpp, broker.cpp, comm.cpp):
// the context behind the opaque mcalf_ctx of include/mcalf_hip.h and the internal functions one file offers the others.
// Plain C++ against the HIP runtime API; kernels are launched through the entry-point table of kernel_args.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types only: the library is resolved at run time (mcalf_comm_*), never linked

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#pragma GCC visibility push(default)     // the C ABI is what the library exports; everything else stays inside it
#include "../../include/mcalf_hip.h"
#pragma GCC visibility pop
#include "kernel_args.h"

#define MCALF_STR_(x) #x
#define MCALF_STR(x) MCALF_STR_(x)

namespace mcalf {
constexpr int kMaxChunks = 8;
constexpr size_t kSmallDoubles = 65536;     // up to 512 KB of parameters (and as many results) go the zero-copy way
constexpr int64_t kStreamTiledMaxItems = 65536;   // tiled spectra stream up to this many work items (host_stream.cpp: run_host_stream)
}  // namespace mcalf
using namespace mcalf;

// A persistent helper thread of a context (the other devices of a multi-device context: host_multi.cpp; the staging copy
// of a large pageable batch: host_abi.cpp).  One job at a time: post(), then wait().  The thread looks for the next job
// for a short while after one -- a sampler calls back to back -- and then sleeps on a condition variable.
typedef int (*WorkFn)(void* who, int64_t lo, int64_t hi, void* arg);
struct HostWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<unsigned> posted{0}, done{0};
    bool quit = false;
    int device = -1;                        // hipSetDevice once, when the thread starts (-1: the thread makes no HIP call)
    void* who = nullptr;
    WorkFn fn = nullptr;
    void* arg = nullptr;
    int64_t lo = 0, hi = 0;
    int rc = 0;

    void start() { th = std::thread([this] { loop(); }); }
    void loop() {
        if (device >= 0) (void)hipSetDevice(device);
        unsigned seen = 0;
        for (;;) {
            bool got = false;
            for (int i = 0; i < 40000 && !got; ++i) {          // 40000 pause instructions (~1 ms) of looking before the thread gives the core up
                got = posted.load(std::memory_order_acquire) != seen;
                if (!got) __builtin_ia32_pause();
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || posted.load(std::memory_order_acquire) != seen; });
                if (posted.load(std::memory_order_acquire) == seen) return;      // quit
            }
            rc = fn(who, lo, hi, arg);
            ++seen;
            done.store(seen, std::memory_order_release);
        }
    }
    void post(WorkFn f, void* a, int64_t l, int64_t h) {
        std::lock_guard<std::mutex> lk(m);                    // (the thread tests its predicate under this lock: no lost wake-up)
        fn = f; arg = a; lo = l; hi = h;
        posted.fetch_add(1, std::memory_order_release);
        cv.notify_one();
    }
    int wait() {
        const unsigned want = posted.load(std::memory_order_relaxed);
        for (unsigned long spins = 0; done.load(std::memory_order_acquire) != want; ++spins) {
            if ((spins & 0xFFFul) == 0xFFFul) std::this_thread::yield();
            else __builtin_ia32_pause();
        }
        return rc;
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_one(); }
        if (th.joinable()) th.join();
    }
};

// Every tunable the ENVIRONMENT of the loading process may set, read ONCE per context at the top of mcalf_create
// (host_config.cpp: the only place the library looks at its environment).  -1 / empty = the variable is not set (or not valid): the
// context then keeps its built-in value.  mcalf_get_config prints the EFFECTIVE values the context runs under, and which
// of them the environment supplied, so a measured number can name its knobs.
struct EnvKnobs {
    int lines_per_sync = -1;                // MCALF_LINES_PER_SYNC   4 / 5
    int persist = -1;                       // MCALF_PERSIST          0 / 1
    int order = -1;                         // MCALF_ORDER            0 / 1
    int inline_max = -1;                    // MCALF_INLINE_MAX       work items
    int resident_us = -1;                   // MCALF_RESIDENT_US      0 .. 1000000
    int setup_block = -1;                   // MCALF_SETUP_BLOCK      64 .. 512, a multiple of 64
    int host_plan[kMaxChunks] = {};         // MCALF_HOST_PLAN        "1,3,4": relative sizes of the pipelined entry's row blocks
    int host_plan_n = 0;
    int host_first_kb = -1;                 // MCALF_HOST_FIRST_KB    bytes (KiB) of the pipelined entry's first row block
    int host_trace = -1;                    // MCALF_HOST_TRACE       0 / 1: host-side time per phase of the pipelined entry
    int stage_threads = -1;                 // MCALF_STAGE_THREADS    helper threads of the staging copy (0: the calling thread alone)
    int stream = -1;                        // MCALF_STREAM           0 / 1 / 2
    int stream_wgs = -1;                    // MCALF_STREAM_WGS
    int stream_min = -1;                    // MCALF_STREAM_MIN       1 .. 64 work items per workgroup slot
    int stream_poll = -1;                   // MCALF_STREAM_POLL      0 / 1
    int stream_eager = -1;                  // MCALF_STREAM_EAGER
    int stream_chunk = -1;                  // MCALF_STREAM_CHUNK     8 .. 512, a multiple of 8
    int stream_device = -1;                 // MCALF_STREAM_DEVICE    0 / 1 / 2
    int stream_trace = -1;                  // MCALF_STREAM_TRACE     0 / 1
    double stream_timeout_s = -1.0;         // MCALF_STREAM_TIMEOUT   seconds, (0, 60]
    int chunks = -1;                        // MCALF_CHUNKS           0 .. 8
    std::string rccl_lib;                   // MCALF_RCCL_LIB         (read when the first mcalf_comm_* call loads RCCL)
#ifdef MCALF_TESTING
    int test_fail_preflight = -1;           // MCALF_TEST_FAIL_PREFLIGHT
    long test_xcd_mask = -1;                // MCALF_TEST_XCD_MASK
    int test_starve = -1;                   // MCALF_TEST_STARVE
#endif
};
EnvKnobs read_environment();                // host_config.cpp

// Diagnostics (MCALF_HOST_TRACE / MCALF_STREAM_TRACE): accumulators of ONE context (a context is not re-entrant, so they need no
// lock; the entries of a multi-device context each keep and print their own), printed when the context is destroyed.
struct HostTrace { double stage_us = 0, stage_bytes = 0, helper_bytes = 0, helper_wait_us = 0, enqueue_us = 0, wait_us = 0, out_us = 0, first_enqueued_us = 0;
                   double call_copy_us = 0, call_order_us = 0, call_launch_us = 0, call_copy_max = 0; long calls = 0, blocks = 0;
                   double small_prep_us = 0, small_launch_us = 0, small_copy_us = 0, small_poll_us = 0, small_out_us = 0; long small_calls = 0; };
struct BlockTimeline { int n = 0; long rows[kMaxChunks]; float h2d0[kMaxChunks], h2d1[kMaxChunks], done[kMaxChunks]; double host_enq[kMaxChunks], call_h2d[kMaxChunks], call_order[kMaxChunks], call_launch[kMaxChunks]; int pinned = 0; double sync_us = 0; };
struct StreamTrace { double t[6] = {}; long n = 0; };

struct mcalf_ctx {
    HostTrace htrace;
    BlockTimeline btrace;
    StreamTrace strace;
    EnvKnobs env;                           // what the environment said when the context was created
    int device = 0;
    std::string arch;
    std::string err;
    // problem
    long npix = 0;
    int nlines = 0, ncompmax = 0, nfill = 0, freespecres = 0, freecont = 0, conv_mode = 0;
    int ndim = 0, startind = 0, endind = 0;
    double specres_fixed = 0, specres_max = 0, contval_fixed = 1, velstep = 0;
    int asymm = 0;
    double veto4 = 0, veto5 = 0;
    // geometry
    int n_cap = 0, tile = 0, ntiles = 0, ncl_cap = 0, jax_half = 0, selfhalo = 0, lps = 4;
    size_t lds_bytes = 0, lds_bytes_inline = 0;    // (the one-launch variant of small calls folds 4 lines per barrier)
    int inline_max_items = 0;                      // launches of at most this many work items take the one-launch variant
    // device buffers
    double *d_nu = nullptr, *d_obj = nullptr, *d_ispec2 = nullptr, *d_lgis = nullptr, *d_err = nullptr, *d_tabs = nullptr;
    LineDev* d_lines = nullptr;
    double* d_wtab = nullptr;
    unsigned long long* d_segok = nullptr;
    double dnu_seg = 0;
    // workspaces (grown on demand)
    double *d_P = nullptr, *d_out = nullptr, *d_partial = nullptr, *d_model = nullptr, *d_bounds = nullptr, *d_prior = nullptr;
    double *d_recs = nullptr, *d_taps = nullptr;
    SampleHdr* d_hdr = nullptr;
    size_t cap_P = 0, cap_out = 0, cap_partial = 0, cap_model = 0, cap_recs = 0, cap_taps = 0, cap_hdr = 0;
    hipStream_t stream = nullptr;
    // optional per-launch timing of the fused kernel (mcalf_profile_begin / _end)
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    bool profiling = false;
    // Small host-pointer calls (the one-theta-at-a-time solvers): parameters and results travel through a
    // page-locked, device-mapped staging block that the kernels read / write directly -- no copy commands.
    // Resident one-theta evaluator (mcalf_set_resident; off by default): its mailbox, its own stream, what the host
    // believes about the kernel, the next request number
    ResidentBox* h_box = nullptr;       // page-locked, coherent, device-mapped
    ResidentBox* d_box = nullptr;
    ResidentShared* d_res_shared = nullptr;   // device words of the evaluator's launch (idle clock, leave flag)
    hipStream_t res_stream = nullptr;
    int resident_us = 0;                // idle limit in microseconds; 0 = no resident kernel
    bool res_alive = false;
    unsigned res_seq = 0;
    long res_launches = 0, res_calls = 0;
    std::vector<double> h_prior;        // the prior box as mcalf_set_prior took it: lo[ndim], hi[ndim] (host copy)
    double* h_small = nullptr;          // host address
    double* d_small = nullptr;          // the same memory as the device sees it
    // prior box of mcalf_set_prior (device copy in d_prior: lo[ndim] then hi[ndim])
    bool prior_set = false;
    int prior_int = 0;
    // Chunked issue: a batch is cut into row blocks that go to the caller's stream and to context-owned
    // auxiliary streams (fork / join through events), so that the set-up kernel and the first workgroups of
    // block k+1 run in the tail of block k.  chunks_req: 0 = automatic, n = exactly n blocks (1 = off).
    int chunks_req = 0;
    int host_plan[kMaxChunks] = {};            // MCALF_HOST_PLAN: relative sizes of the row blocks of the pipelined
    int host_plan_n = 0;                       // host-pointer entry (0 = the built-in plan: first block by BYTES, then doubling)
    int host_first_kb = 128;                   // MCALF_HOST_FIRST_KB: parameter bytes (KiB) of the pipelined entry's first row block
    int host_trace = 0;                        // MCALF_HOST_TRACE=1 (diagnostic): host-side time per phase of the pipelined entry
    int stage_threads = 0;                     // MCALF_STAGE_THREADS: helper threads of the pageable -> page-locked staging copy
    std::vector<std::unique_ptr<HostWorker>> stagers;   // (started by the first call that uses them)
    int num_cu = 256;
    int persist = 1;                    // fused kernel as a persistent grid (MCALF_PERSIST=0: one workgroup per item)
    int setup_block = 512;              // threads per workgroup of the set-up kernel (MCALF_SETUP_BLOCK: 64 .. 512)
    unsigned int* d_queue = nullptr;    // [kMaxChunks] work-item queues of the persistent kernel
    int* d_order = nullptr;             // [batch] hand-out order of the persistent kernel (per row block)
    size_t cap_order = 0;
    int ordered = 1;                    // MCALF_ORDER=0: hand the live points out in row order
    hipStream_t aux[kMaxChunks - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxChunks - 1] = {};
    hipEvent_t ev_h2d[kMaxChunks] = {};     // row-block pipeline: block k's rows are in HBM (recorded on the copy stream, aux[1])
    // multi-GPU: the communicator of mcalf_comm_init (one process per GPU, RCCL over xGMI)
    ncclComm_t comm = nullptr;
    int comm_ranks = 0, comm_rank = -1;
    hipStream_t comm_stream = nullptr;      // the exchange runs here, behind an event of the launch stream
    hipEvent_t ev_kernels = nullptr;        // launch stream -> comm stream: this step's logL block is complete
    hipEvent_t ev_comm[2] = {};             // comm stream -> launch stream: exchange of call k (slot k & 1) has landed
    bool ev_comm_used[2] = {};
    unsigned comm_calls = 0;
    int comm_overlap = 0;                   // 0: every gather call ends with the launch stream waiting for its exchange
    bool comm_dead = false;                 // aborted after a failure inside an exchange
#ifdef MCALF_TESTING
    bool fail_preflight = false;            // MCALF_TEST_FAIL_PREFLIGHT=1 (test builds only): an injected workspace-growth failure
#endif
    // page-locked staging of the host-pointer entries: parameter rows in, scalars out
    double* h_stage = nullptr;
    size_t cap_stage = 0;
    // Streaming single launch of the host-pointer entries (run_host_stream): queues / stamps in HBM, the words the host
    // and the kernel exchange in a page-locked, device-mapped block (h_ctl: [0] status, [1] generation of the last
    // launch that has drained, [16] rows staged so far -- a cache line of its own)
    // (ONE allocation: records, taps, parameter rows, headers, stamps of `cap_sws` live points, then the queues)
    void* d_sws = nullptr;
    size_t cap_sws = 0;
    double *s_recs = nullptr, *s_taps = nullptr, *s_P = nullptr;
    size_t s_rec_stride = 0, s_tap_stride = 0, s_hdr_stride = 0;
    SampleHdr* s_hdr = nullptr;
    StreamCtl* d_sctl = nullptr;
    unsigned int* d_ready = nullptr;
    volatile unsigned int* h_ctl = nullptr;
    unsigned int* d_ctl = nullptr;          // the same words as the device sees them
    unsigned int stream_gen = 0;            // stamp of the last streaming launch; never restarts (a regrown workspace is zero-filled, and
                                            // a restarted count could meet the completion word of an earlier launch: h_ctl[1])
    // Which XCDs the context's stream reaches: bit i = the probe kernel saw a workgroup with hardware XCC_ID i
    // (stream_probe_xcds: mcalf_create, mcalf_set_cu_mask).  The streaming launch is built for exactly 0xFF.
    unsigned int xcd_mask = 0;
    std::vector<uint32_t> cu_mask;          // mcalf_set_cu_mask: the CU mask of the context's own streams (empty: none)
    int stream_on = 1;                      // MCALF_STREAM: 0 = the row-block pipeline of round 2 instead; 1 = automatic (spectra that fit
                                            // one tile always, tiled ones up to kStreamTiledMaxItems work items: measured, config E's
                                            // 16384 x 5 items run 0.5 % faster through the pipeline, 8192 x 5 items 3 % slower); 2 = always
    int stream_min = 4;                     // MCALF_STREAM_MIN: work items per workgroup slot from which a host-pointer batch streams
    int stream_wgs = 16;                    // MCALF_STREAM_WGS: workgroups dedicated to the set-up while rows are outstanding
    int stream_eager = 0;                   // MCALF_STREAM_EAGER: blocks of 8 rows per XCD any workgroup may set up (0: what the first items need)
    int stream_chunk = 32;                  // MCALF_STREAM_CHUNK: rows such a workgroup claims (and copies to HBM) at a time
    int stream_trace = 0;                   // MCALF_STREAM_TRACE=1 (diagnostic): host-side time per phase of the streaming entry
    int stream_device = 0;                  // MCALF_STREAM_DEVICE=1 (diagnostic): the *_device scalar entries take the streaming launch too
    int stream_poll = 1;                    // MCALF_STREAM_POLL=0: wait for the stream's signal instead of polling h_ctl[1]
    double stream_timeout_s = 0.5;          // MCALF_STREAM_TIMEOUT: longest wait of a wave inside the kernel
    // LSF wider than a workgroup tile (numpy boundary): the fused kernel runs WITHOUT convolution and continuum into
    // d_wide, two more kernels do taps, periodic convolution, continuum and terms from HBM (host_abi.cpp: launch_wide).
    // n_cap is 0 for such a context (the tile carries no halo); wide_n_cap is the half-width provisioned from specres_max.
    int wide = 0, wide_n_cap = 0;
    double *d_wide = nullptr, *d_wtaps = nullptr, *d_wpartial = nullptr, *d_wrows = nullptr;
    SampleHdr* d_whdr = nullptr;
    size_t cap_wide = 0, cap_wtaps = 0, cap_wpartial = 0, cap_wrows = 0, cap_whdr = 0;
    mcalf_launch_info_t last = {};      // what the last call did (mcalf_last_launch)
    // Single-process multi-device context (mcalf_create_multi, host_multi.cpp): the parent holds one complete context per
    // device entry and a worker thread for each but the first; it owns no device memory itself (its problem / geometry
    // fields are copies of sub-context 0's, for mcalf_info).
    std::vector<mcalf_ctx*> subs;
    struct MultiPool* pool = nullptr;
    int multi_last_active = 0;              // sub-contexts the last call was cut over
};
inline bool is_multi(const mcalf_ctx* ctx) { return !ctx->subs.empty(); }
constexpr int kCtlWords = 64, kCtlArrived = 16, kCtlFinalized = 4;   // ([4]: stamp of the last streaming launch whose FINALIZE kernel has finished)

// ---- host_abi.cpp ---------------------------------------------------------------------------------------------------
int set_err(mcalf_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
void set_last_error(const std::string& msg);          // the thread's message (mcalf_last_error(NULL))

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return set_err(ctx, MCALF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

template <typename T>
inline int grow(mcalf_ctx* ctx, T** ptr, size_t* cap, size_t need_elems) {
    if (need_elems <= *cap) return MCALF_OK;
    if (*ptr) HIP_TRY(ctx, hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)ptr, need_elems * sizeof(T)));
    *cap = need_elems;
    return MCALF_OK;
}

inline double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// The kernel arguments of rows [row0, row0 + nrows) of a batch (everything but the launch geometry).
// wide_stage (contexts whose LSF is wider than a tile, launch_wide): which of the launch's kernels the block is for
enum { kWideNone = 0, kWideFused = 1, kWideKernels = 2 };
KArgs make_kargs(const mcalf_ctx* ctx, int mode, const double* dP, int64_t row0, int64_t nrows, int chunk, int targonly,
                 int onecomp_fill, double* d_out, double* d_model, bool from_cube, double* d_theta, int wide_stage = kWideNone);
// Enqueue one batch on `stream` (asynchronous): set-up kernel, fused kernel, finalize when tiled.
int launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, int targonly, int onecomp_fill, double* d_out,
           double* d_model, hipStream_t stream, bool from_cube = false, double* d_theta = nullptr);
// Everything of a launch that can fail WITHOUT anything having been enqueued (range check, workspace growth).
int launch_preflight(mcalf_ctx* ctx, int mode, int64_t batch);
int launch_finalize(mcalf_ctx* ctx, const KArgs& a, int64_t nrows, int mode, hipStream_t stream, int nparts = 0, bool signal = false);
int ensure_small(mcalf_ctx* ctx);                      // the page-locked block of small calls
bool is_pinned_host(const void* p);
// A stream of the context: created with the context's CU mask when it has one (mcalf_set_cu_mask).
int create_stream(mcalf_ctx* ctx, hipStream_t* out, int priority = 0);

void apply_environment(mcalf_ctx* ctx);                // host_config.cpp: the environment's snapshot over the built-in values
// Entries that make no sense on a multi-device parent (device pointers belong to ONE device; a communicator, a resident
// kernel, a profile to one context) refuse it.
#define MCALF_SINGLE_ONLY(ctx, what)                                                                              \
    do {                                                                                                          \
        if ((ctx) && is_multi(ctx))                                                                               \
            return set_err(ctx, MCALF_ERR_INVALID, "%s: not available on a multi-device context (mcalf_create_multi); " \
                           "use a single-device context per GPU", what);                                          \
    } while (0)

// ---- host_multi.cpp: one process driving several devices ---------------------------------------------------------------
void multi_release(mcalf_ctx* ctx);                    // stops the workers, destroys the sub-contexts
// Rows [lo, hi) of `batch` that sub-context k of n evaluates (contiguous blocks; the arithmetic of mc-alf_amd/dist.py).
inline void multi_bounds(int64_t batch, int n, int k, int64_t* lo, int64_t* hi) {     // (shard_bounds of dist.py: sizes differ by at most one)
    const int64_t base = batch / n, extra = batch % n;
    *lo = (int64_t)k * base + std::min<int64_t>(k, extra);
    *hi = *lo + base + (k < extra ? 1 : 0);
}
int multi_active(const mcalf_ctx* ctx, int64_t batch);  // sub-contexts a batch of this size is cut over (>= 1)
// fn(sub, k, lo, hi) on every active sub-context concurrently (sub 0 on the calling thread); first error wins.
int multi_run(mcalf_ctx* ctx, int64_t batch, WorkFn fn, void* arg);      // (fn's `who` is the sub-context)

// ---- host_stream.cpp: ONE streaming launch for a large host-pointer batch ----------------------------------------------
int stream_probe_xcds(mcalf_ctx* ctx);                 // which XCDs the context's stream reaches (ctx->xcd_mask)
int stream_prepare(mcalf_ctx* ctx, int mode, int64_t batch);
int ensure_ctl(mcalf_ctx* ctx);                        // the page-locked words a launch and the host exchange (h_ctl / d_ctl)
int stream_launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, double* d_out, hipStream_t stream, int wgs,
                  int64_t eager_rows, bool staged, bool host_rows, bool from_cube = false);
bool stream_qualifies(const mcalf_ctx* ctx, int64_t batch);
void host_scale_cube(const mcalf_ctx* ctx, const double* cube, int64_t batch, double* theta);
int run_host_stream(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, double* out_scalar, bool* taken,
                    bool from_cube = false, double* theta_out = nullptr);
void stream_trace_report(mcalf_ctx* ctx);
void host_trace_report(mcalf_ctx* ctx);          // host_abi.cpp: MCALF_HOST_TRACE, the row-block pipeline's phases

// ---- broker.cpp: resident one-theta evaluator ---------------------------------------------------------------------------
void resident_stop(mcalf_ctx* ctx);
bool resident_serves(const mcalf_ctx* ctx, int mode, int64_t batch, int rowlen, bool from_cube);
int resident_call(mcalf_ctx* ctx, const double* row, int rowlen, double* out);

// ---- comm.cpp -----------------------------------------------------------------------------------------------------------
void comm_release(mcalf_ctx* ctx);
