/* mcalf_hip.h -- C ABI of libmcalf_hip.so: the MI355X (gfx950) implementation of the
 * MC-ALF likelihood hot path (Voigt synthesis -> exp(-tau) -> Gaussian-LSF convolution
 * -> Gaussian log-likelihood) for a batch of live points.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, never throws, and
 * returns 0 (MCALF_OK) or a negative MCALF_ERR_* code; the message is available from
 * mcalf_last_error().  Per-sample numerical failures are VALUES in the output
 * (nan / -inf), not errors -- as in the reference, where invalid models surface through
 * np.nansum (hires_fitter.py:294).
 *
 * Reference interfaces replaced (paths relative to /root/reference/mcalf/routines/):
 *   mcalf_create           <- als_fitter.__init__ data/layout state   hires_fitter.py:32-200
 *   mcalf_loglike_batch    <- als_fitter.lnlhood_worker (per row)     hires_fitter.py:287-328
 *                             (and the closure of get_jax_likelihood  hires_fitter.py:685-693
 *                              when conv_mode = MCALF_CONV_SAME_EDGE_JAX)
 *   mcalf_model_batch      <- als_fitter.reconstruct_spec(p,targonly) hires_fitter.py:409-449
 *   mcalf_onecomp_batch    <- reconstruct_onecomp / _onecomp_fill     hires_fitter.py:379-406
 *   mcalf_chi2_batch       <- als_fitter.chi2                         hires_fitter.py:236-248
 *   mcalf_scale_cube_batch <- _scale_cube_pc / _scale_cube_mn         hires_fitter.py:202-216
 *   mcalf_loglike_cube_batch <- lnlhood_pc(_scale_cube_pc(cube))      hires_fitter.py:202-209,250-262
 *   mcalf_voigt_hjerting[_nodes] <- scipy.special.wofz(u + i a).real  hires_fitter.py:365
 *                             / voigt_jax.hjert                       voigt_jax.py:121-127
 *   mcalf_set_cu_mask, mcalf_stream_partition
 *                          <- the solver hands over host arrays and trusts the float it gets back
 *                             (hires_fitter.py:250-262,287-294): the host-pointer entries must be right on ANY
 *                             gfx950 device shape (partition modes, CU masks), not only the one they are fastest on
 *   mcalf_set_resident     <- the solvers' one-theta calling pattern  hires_fitter.py:250-285
 *                             (lnlhood_pc / _dy / _mn, one call per proposed point)
 *   mcalf_broker_serve[_resident], mcalf_mailbox_call
 *                          <- one solver rank per core, each calling  ../cli.py:37-41,110
 *                             the likelihood serially (PolyChord's MPI workers)
 *   mcalf_comm_* / mcalf_loglike_gather[v]_device
 *                          <- data parallelism over live points       ../cli.py:110,274-280
 *   mcalf_create_multi, mcalf_last_launch_sub
 *                          <- ONE process holds the whole batch       ../cli.py:274-280 (jaxns vmaps the likelihood over
 *                             the live points inside one Python process): one context that drives several devices
 *   mcalf_get_config       <- (none: the reference has no tunables on this path; the library says which of its own
 *                             the loading process's environment has set)
 *
 * Ownership: the context owns all device memory it allocates.  Host pointers passed to
 * any call are borrowed for the duration of that call only.  `*_device` entries take
 * DEVICE pointers valid on the context's GPU and enqueue on the given hipStream_t
 * (passed as void*; NULL = the legacy default stream) without synchronising.
 *
 * Threading: a context is not re-entrant (one in-flight call per context, and every
 * asynchronous call of a context on the same stream); distinct contexts are independent.
 * A resolution beyond the provisioned specres_max makes a row's model NaN, its logL -inf and
 * its chi2 +inf (the reference would allocate a longer kernel).  The provisioned LSF itself may be of
 * ANY width, under both boundary modes: one whose halo does not fit a 4096-pixel workgroup tile is
 * convolved by a second pair of kernels behind the fused one (same entries, same conventions;
 * hires_fitter.py:458-464 / the fixed grid of :549-560).  Only a MCALF_CONV_SAME_EDGE_JAX grid longer than
 * the spectrum is refused (MCALF_ERR_INVALID): the reference's jnp.convolve / jnp.where cannot broadcast
 * it either.
 */
#ifndef MCALF_HIP_H
#define MCALF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCALF_ABI_VERSION 7   /* 2: mcalf_last_launch, gatherv / overlap / join, version string carries the source hash
                                 3: MCALF_PATH_HOST_STREAM, mcalf_launch_info_t grows by stream_setup_wgs / stream_polled
                                 4: mcalf_broker_serve
                                 5: mcalf_set_resident, mcalf_broker_serve_resident
                                 6: mcalf_set_cu_mask, mcalf_stream_partition, mcalf_launch_info_t grows by xcd_mask / stream_wgs_min /
                                    stream_wgs_max / stream_fallback (the streaming launch is taken only on the device shape it
                                    was built for, and checked after every launch)
                                 7: mcalf_create_multi, mcalf_last_launch_sub, mcalf_get_config; mcalf_info_t grows by ndevices /
                                    devices[16], mcalf_launch_info_t by devices_used */

enum {
    MCALF_OK = 0,
    MCALF_ERR_INVALID = -1,   /* bad argument / inconsistent spec          */
    MCALF_ERR_HIP = -2,       /* a HIP runtime call failed                 */
    MCALF_ERR_NODEVICE = -3,  /* no usable gfx950 device                   */
    MCALF_ERR_RANGE = -4,     /* problem exceeds a build-time capacity     */
    MCALF_ERR_NOMEM = -5,
    MCALF_ERR_COMM = -6       /* RCCL missing or a collective call failed  */
};

enum {
    MCALF_CONV_WRAP_NUMPY = 0,    /* numpy path: periodic boundary, per-sample kernel length,
                                     convolution skipped when R <= velstep (hires_fitter.py:445-464) */
    MCALF_CONV_SAME_EDGE_JAX = 1  /* JAX path: fixed kernel grid from max specres, zero padding,
                                     edges reset to the unconvolved model, always convolved,
                                     floor() on the ncomp slot (hires_fitter.py:549-560,616,667-681) */
};

typedef struct mcalf_ctx mcalf_ctx;

/* One transition: rest wavelength [Angstrom], oscillator strength, damping gamma [1/s]. */
typedef struct {
    double wrest_A;
    double f;
    double gamma;
} mcalf_line;

/* Problem definition: what als_fitter.__init__ leaves in `self` for the likelihood to read. */
typedef struct {
    int64_t npix;            /* number of selected pixels                                    */
    const double* wl;        /* [npix] obj_wl, Angstrom            (host, copied)            */
    const double* flux;      /* [npix] obj                         (host, copied)            */
    const double* err;       /* [npix] obj_noise                   (host, copied)            */
    double velstep;          /* km/s per pixel (hires_fitter.py:84-87)                       */
    int32_t nlines;          /* numlines                                                     */
    const mcalf_line* lines; /* [nlines] linepars                  (host, copied)            */
    mcalf_line fill;         /* linefill (hires_fitter.py:120-121)                           */
    int32_t ncompmax;        /* upper bound of the ncomp slot                                */
    int32_t nfill;           /* number of filler components                                  */
    int32_t freespecres;     /* 1: p[0] is the LSF FWHM                                      */
    int32_t freecont;        /* 1: continuum is a free parameter                             */
    double specres_fixed;    /* FWHM km/s used when !freespecres: max(specres) for the numpy
                                path, specres[0] for the JAX path                            */
    double specres_max;      /* largest FWHM any sample may carry (sizes the LDS halo and the
                                fixed JAX kernel grid)                                       */
    double contval_fixed;    /* continuum when !freecont                                     */
    int32_t conv_mode;       /* MCALF_CONV_*                                                 */
    int32_t device;          /* HIP device ordinal, or -1 for the current device             */
    int32_t asymmlike;       /* 1: asymmetric veto of hires_fitter.py:296-303 in loglike      */
    double asymm_n4;         /* gauss_cdf[1]: allowed count of (obj-model)/err > 4 before the
                                0.01*npix grace (the reference draws it from an unseeded
                                np.random.normal, :179-181, so it is an input here)          */
    double asymm_n5;         /* gauss_cdf[2]: same for > 5                                    */
} mcalf_spec;

typedef struct {
    int32_t abi_version;
    int32_t ndim;       /* [freespecres] + [freecont] + 1 + 3*ncompmax + 3*nfill             */
    int32_t startind;   /* index of the ncomp slot      (hires_fitter.py:169-174)            */
    int32_t endind;     /* first filler parameter       (hires_fitter.py:176)                */
    int32_t n_cap;      /* LSF half-width (pixels) provisioned from specres_max              */
    int32_t tile;       /* pixels per workgroup tile                                         */
    int32_t ntiles;     /* tiles per sample                                                  */
    int32_t device;     /* HIP device ordinal in use                                         */
    int64_t npix;
    char arch[32];      /* gcnArchName of the device                                         */
    int32_t ndevices;   /* device entries of the context: 1, or what mcalf_create_multi took */
    int32_t devices[16];/* their HIP ordinals (entries may repeat)                           */
} mcalf_info_t;

/* Create / destroy.  mcalf_create never returns a half-built context: on failure *out is
 * NULL and mcalf_last_error(NULL) holds the message. */
int mcalf_create(const mcalf_spec* spec, mcalf_ctx** out);
/* ONE context over several devices of this process (spec->device is ignored; `devices` lists 1 .. 16 HIP ordinals, an
 * ordinal may repeat -- two entries on one GPU are two independent sub-contexts).  The reference's large batches arise
 * inside one process (jaxns vmaps the likelihood over the live points, cli.py:274-280): the host-pointer entries of such
 * a context -- mcalf_loglike_batch, _chi2_batch, _model_batch, _onecomp_batch, _loglike_cube_batch -- cut the rows into
 * contiguous blocks whose sizes differ by at most one (entry k of the n entries in use; an entry gets at least 256
 * rows, so small calls and the one-theta callables run on entry 0 alone), issue every device's call concurrently -- the
 * calling thread drives entry 0, a helper thread of the context each of the others -- and every device writes its block of
 * results straight into the caller's array: no collective, no device-to-device traffic.  A live point's arithmetic does
 * not depend on the shard: the results equal the single-device ones bit for bit.  mcalf_set_prior / _set_chunks /
 * _set_cu_mask / _reserve apply to every entry, mcalf_set_resident to entry 0; the *_device entries, the profile brackets,
 * the broker and the mcalf_comm_* family need a single-device context (MCALF_ERR_INVALID). */
int mcalf_create_multi(const mcalf_spec* spec, const int32_t* devices, int32_t ndevices, mcalf_ctx** out);
/* How a multi-device context of `nentries` device entries cuts a batch: *entries_used = the entries that take part (every one
 * of them gets at least 256 rows), [*lo, *hi) = the rows of entry k (empty for an entry that sits the call out).  Pure host
 * arithmetic, no device needed -- the functions mcalf_create_multi's contexts use; block sizes differ by at most one, the
 * first batch % entries_used entries take the longer ones (mc-alf_amd/dist.py::shard_bounds, the split of the one-process-
 * per-GPU form). */
int mcalf_shard_bounds(int64_t batch, int32_t nentries, int32_t k, int32_t* entries_used, int64_t* lo, int64_t* hi);
void mcalf_destroy(mcalf_ctx* ctx);
int mcalf_info(const mcalf_ctx* ctx, mcalf_info_t* info);
const char* mcalf_last_error(const mcalf_ctx* ctx);
/* "mcalf_hip <version> (gfx950, abi <n>) src <hash>": <hash> = first 16 hex digits of the sha256 over the kernel
 * sources the library was built from ("unstamped" for a build that did not go through mc-alf_amd/build.py). */
const char* mcalf_version(void);
/* The configuration the context runs under, as text ("name=value ..." followed by "[env: ...]", the MCALF_* variables that
 * were set and valid when the context was created -- the library reads its environment exactly once per context, in
 * mcalf_create): written into buf (n bytes, always terminated; ~600 bytes suffice). */
int mcalf_get_config(const mcalf_ctx* ctx, char* buf, int64_t n);

/* Pre-size the context's device workspaces for batches up to `batch` rows so that later
 * calls (including *_device calls captured into a hipGraph) allocate nothing. */
int mcalf_reserve(mcalf_ctx* ctx, int64_t batch);

/* logL[i] = lnlhood_worker(P[i, :]) for i < batch.   P row-major [batch][ndim], host. */
int mcalf_loglike_batch(mcalf_ctx* ctx, const double* P, int64_t batch, double* logL);
/* flux[i, :] = reconstruct_spec(P[i, :], targonly).  flux row-major [batch][npix], host. */
int mcalf_model_batch(mcalf_ctx* ctx, const double* P, int64_t batch, int32_t targonly, double* flux);
/* chi2[i] = nansum(ispec2 (obj - model)^2); +inf when the model is identically zero. */
int mcalf_chi2_batch(mcalf_ctx* ctx, const double* P, int64_t batch, double* chi2);
/* Single-component spectra: rows of Q are (R, cont, N, z, b).  which = 0: every line of the
 * component (reconstruct_onecomp); 1: the filler line (reconstruct_onecomp_fill); 2 + k: line k
 * alone (the per-line model calc_w integrates, hires_fitter.py:483). */
int mcalf_onecomp_batch(mcalf_ctx* ctx, const double* Q, int64_t batch, int32_t which, double* flux);

/* Resident one-theta evaluator (off by default).  The solvers call the likelihood one theta at a time (lnlhood_pc / _dy /
 * _mn: hires_fitter.py:250-285); of such a call about a third is the kernel LAUNCH, and launches of different processes
 * serialise.  With idle_us > 0, mcalf_loglike_batch calls of ONE row on a spectrum that fits one pixel tile are answered by
 * one workgroup that STAYS on the chip between calls and takes its requests from a page-locked mailbox -- same arithmetic,
 * same bits, no launch.  The kernel leaves by itself after idle_us microseconds without a request (the next call starts
 * another one), so a device-wide synchronisation never waits longer than that; idle_us = 0 tells it to leave now and turns
 * the mode off.  Other entries of the context are not affected (the evaluator has its own stream and its own memory).
 * The environment variable MCALF_RESIDENT_US gives the initial value. */
int mcalf_set_resident(mcalf_ctx* ctx, int32_t idle_us);

/* Likelihood broker, serving side: MANY one-theta-at-a-time solver ranks (PolyChord runs one MPI rank per core, each calling
 * lnlhood_pc(theta) serially: cli.py:37-41, 110; hires_fitter.py:250-262) served by ONE thread of ONE process that owns the
 * device contexts.  The request block lives wherever the caller puts it (mc-alf_amd/broker.py: POSIX shared memory); rank s
 * writes its row theta[s * theta_stride ..], then bumps req[s * counter_stride]; the server evaluates every open request of
 * the moment as one small batch on a context that is free, writes logl[s * logl_stride] and then sets ack[..] to the
 * request number.  With several contexts, requests that arrive while a launch is in flight leave at once on the next free
 * context.  Returns when *stop becomes non-zero (after answering what is in flight) or after max_seconds (0: never). */
typedef struct {
    int32_t slots;              /* request slots, one per solver rank */
    int32_t ndim;               /* doubles per parameter row (the contexts' ndim) */
    volatile uint64_t* req;     /* [slots * counter_stride] */
    volatile uint64_t* ack;     /* [slots * counter_stride] */
    int64_t counter_stride;     /* in 64-bit words (8: one cache line per slot) */
    const double* theta;        /* [slots * theta_stride] */
    int64_t theta_stride;       /* in doubles, >= ndim */
    double* logl;               /* [slots * logl_stride] */
    int64_t logl_stride;        /* in doubles */
    volatile uint64_t* stop;
    uint64_t* stats;            /* optional [2]: launches and thetas served so far (incremented) */
    double idle_sleep_after_s;  /* spin this long without a request before yielding the core between polls */
} mcalf_broker_t;
int mcalf_broker_serve(mcalf_ctx* const* ctxs, int32_t nctx, const mcalf_broker_t* b, double max_seconds);

/* The broker with RESIDENT evaluators: every solver rank gets a workgroup that stays on the chip and takes the rank's requests
 * straight from its mailbox in the shared block -- no server thread and no launch on a call's path.  `boxes` points to `slots`
 * mailboxes of MCALF_MAILBOX_BYTES each (64-byte aligned, zero-filled by whoever creates the block), laid out as
 *   uint32 req, quit, ack, state;  double result;  double reserved[5];  double row[64]
 * Rank s: result <- the bit pattern MCALF_RESULT_PENDING, row <- theta, then req <- req + 1 (in that order; a release store);
 * spin until result differs from the pattern.  This call page-locks the block (hipHostRegister) and keeps ONE launch of
 * `slots` workgroups alive while requests come (workgroup k polls mailbox k; they leave TOGETHER after idle_us without a
 * request on any mailbox, and the next request starts the launch again -- one launch on one stream, because hardware queues
 * are few); it returns when *stop becomes non-zero (every workgroup has left by then) or after max_seconds (0: never).  Serves spectra of one pixel tile with at
 * most 64 parameters (MCALF_ERR_RANGE otherwise: use mcalf_broker_serve).
 * Co-residency: a mailbox is served only while its workgroup is on the chip, and all `slots` workgroups of the launch are
 * expected to be there at once -- one per compute unit, so `slots` may not exceed the device's CU count (MCALF_ERR_INVALID;
 * 256 on an unpartitioned MI355X).  While another kernel occupies compute units (a batch call of this process, another
 * process), workgroups that find no CU leave their mailboxes unpolled until one frees up: such a call waits, it is not lost. */
#define MCALF_MAILBOX_BYTES 576
#define MCALF_RESULT_PENDING 0x7FF8C0DEC0DE0001ull
int mcalf_broker_serve_resident(mcalf_ctx* ctx, void* boxes, int32_t slots, volatile uint64_t* stop, int32_t idle_us,
                                uint64_t* stats, double max_seconds);

/* A rank's side of the mailbox protocol, for solvers written in C / C++ / Fortran (header-only; the Python ranks do the same in
 * mc-alf_amd/broker.py): logL of theta[0 .. ndim) through the mailbox at `box`.  Spins until the rank's workgroup has answered
 * (the result slot no longer holds the pending pattern, or -- should the answer itself be that pattern -- `ack` equals the
 * request number); returns the canonical NaN when *stop (may be NULL) becomes non-zero first, or when max_spins (0: no limit)
 * looks have passed without an answer: a server that has died raises no stop flag, so a caller that cannot watch the server
 * process should pass a limit (a look is ~10 ns; 3e9 is about half a minute). */
#if defined(__GNUC__) || defined(__clang__)
static inline double mcalf_mailbox_call_bounded(void* box, const double* theta, int32_t ndim, const volatile uint64_t* stop,
                                                uint64_t max_spins) {
    volatile uint32_t* words = (volatile uint32_t*)box;            /* req, quit, ack, state */
    volatile uint64_t* result = (volatile uint64_t*)((char*)box + 16);
    double* row = (double*)((char*)box + 64);
    union { uint64_t u; double d; } v;
    int32_t i;
    uint64_t spins = 0;
    const uint32_t seq = words[0] + 1u;
    *result = MCALF_RESULT_PENDING;
    for (i = 0; i < ndim; ++i) row[i] = theta[i];
    __atomic_store_n(&words[0], seq, __ATOMIC_RELEASE);            /* the request number, last */
    while ((v.u = __atomic_load_n(result, __ATOMIC_ACQUIRE)) == MCALF_RESULT_PENDING) {
        if (__atomic_load_n(&words[2], __ATOMIC_ACQUIRE) == seq) { v.u = __atomic_load_n(result, __ATOMIC_ACQUIRE); break; }
        ++spins;
        if ((stop && (spins & 0xFFFFu) == 0 && *stop) || (max_spins && spins >= max_spins)) { v.u = 0x7FF8000000000000ull; break; }
    }
    return v.d;
}
static inline double mcalf_mailbox_call(void* box, const double* theta, int32_t ndim, const volatile uint64_t* stop) {
    return mcalf_mailbox_call_bounded(box, theta, ndim, stop, 0);
}
#endif

/* Row blocks a batch is issued in (0 = automatic [default], n <= 8 = exactly n).  Automatic means ONE block
 * for the *_device entries (the persistent fused kernel leaves no launch tail worth filling; measured) and, for
 * the host-pointer entries with large batches, ONE streaming launch (MCALF_PATH_HOST_STREAM: spectra that fit one pixel
 * tile, and tiled ones up to 65536 work items -- live points x tiles; MCALF_STREAM=2 streams larger tiled batches too; an
 * explicit block count, MCALF_STREAM=0, a larger tiled batch or a launch below the persistent-grid threshold selects the
 * row-block pipeline instead: a first block of
 * 128 KiB of parameter rows (MCALF_HOST_FIRST_KB; twice that for page-locked input), every following block twice the one
 * before, the last one taking the rest, so that the GPU starts after ~10 us of staging and block k+1's staging copy,
 * H2D copy and per-sample set-up run under block k's kernel).  With more than one block in a *_device call the
 * blocks after the first run on context-owned streams between a fork event recorded on the caller's stream and
 * join events that stream waits for: the call keeps plain stream semantics (and can be captured into a
 * hipGraph).  Results do not depend on the setting (every live point is evaluated independently).  The
 * environment variable MCALF_CHUNKS gives the initial value. */
int mcalf_set_chunks(mcalf_ctx* ctx, int32_t nchunks);
/* Row blocks a *_device call of `batch` rows is issued in under the current setting. */
int32_t mcalf_get_chunks(const mcalf_ctx* ctx, int64_t batch);

/* Same, device pointers + stream, asynchronous.
 * Stream rule: all asynchronous calls of ONE context must be issued on ONE stream (the context's per-sample
 * workspaces are shared by its launches and ordered only by that stream); call mcalf_reserve(batch) first when
 * the call is to be captured into a hipGraph or must not allocate (a *_device call otherwise grows the
 * workspaces on first use, which synchronises the device). */
int mcalf_loglike_batch_device(mcalf_ctx* ctx, const double* dP, int64_t batch, double* dlogL, void* stream);
int mcalf_model_batch_device(mcalf_ctx* ctx, const double* dP, int64_t batch, int32_t targonly,
                             double* dflux, void* stream);

/* What the LAST call of this context actually did (tests and benchmarks assert on the path taken instead of
 * inferring it from batch sizes).  `path`: which entry plan ran; the remaining fields describe the last fused-kernel
 * launch of that call. */
enum {
    MCALF_PATH_NONE = 0,          /* no call yet                                                              */
    MCALF_PATH_DEVICE = 1,        /* a *_device entry (caller's device pointers and stream)                   */
    MCALF_PATH_HOST_ZEROCOPY = 2, /* host pointers, small call: theta / logL through the page-locked mapped block */
    MCALF_PATH_HOST_PIPELINED = 3,/* host pointers, large scalar-output call: row blocks over two streams     */
    MCALF_PATH_HOST_STAGED = 4,   /* host pointers, model output: H2D, launch, D2H on the context's stream    */
    MCALF_PATH_HOST_STREAM = 5    /* host pointers, large scalar-output call: ONE streaming launch, no copy commands --
                                     the grid reads the parameter rows from page-locked memory while the host is still
                                     staging them, sets the live points up itself and writes logL into page-locked memory */
};
typedef struct {
    int32_t path;           /* MCALF_PATH_*                                                                   */
    int32_t row_blocks;     /* row blocks the call was issued in                                              */
    int32_t persistent;     /* 1: the last fused launch ran the persistent grid with the work-item queue      */
    int32_t grid;           /* workgroups of the last fused launch                                            */
    int64_t items;          /* work items (live points x tiles) of the last fused launch                      */
    int32_t lines_per_sync; /* lines folded per workgroup barrier (4 or 5)                                    */
    int32_t selfhalo;       /* 1: single-tile spectrum, halo entries are copies of the tile's own pixels      */
    int32_t pinned_in;      /* host-pointer entries: the parameter rows were page-locked caller memory        */
    int32_t pinned_out;     /* host-pointer entries: the result array was page-locked caller memory           */
    int32_t inline_setup;   /* 1: small launch -- ONE kernel, the per-sample set-up ran inside the fused kernel;
                               3: no launch at all -- the context's resident evaluator answered (mcalf_set_resident) */
    int32_t ordered;        /* 1: the persistent grid handed the live points out sorted by component count      */
    int32_t stream_setup_wgs; /* MCALF_PATH_HOST_STREAM: workgroups of the grid dedicated to the set-up while rows were outstanding */
    int32_t stream_polled;  /* MCALF_PATH_HOST_STREAM: 1 = completion seen through the kernel's page-locked word, 0 = stream signal;
                               MCALF_PATH_HOST_ZEROCOPY: 1 = completion read off the results in page-locked memory */
    int32_t xcd_mask;       /* bit i: the context's probe kernel saw workgroups of its stream on XCD i (hardware XCC_ID).  The
                               streaming launch deals rows to XCDs 0 .. 7 and is taken only when this is exactly 0xFF */
    int32_t stream_wgs_min; /* MCALF_PATH_HOST_STREAM (and a launch discarded as STARVED): the fewest / most workgroups of the */
    int32_t stream_wgs_max; /* launch an XCD received, counted by the kernel -- every XCD's rows are evaluated by ITS workgroups only */
    int32_t stream_fallback;/* host-pointer entries, what kept the call from being ONE streaming launch although its size asked
                               for one: 0 nothing (or not applicable), MCALF_STREAM_FALLBACK_* otherwise -- the row-block
                               pipeline (MCALF_PATH_HOST_PIPELINED) evaluated the call instead, same bits */
    int32_t devices_used;   /* device entries the call was cut over (1 for a single-device context)                   */
} mcalf_launch_info_t;
enum {
    MCALF_STREAM_FALLBACK_SHAPE = 1,     /* the stream does not reach exactly the eight XCDs of an unpartitioned MI355X
                                            (DPX / QPX / CPX partition, CU mask): decided before anything was launched */
    MCALF_STREAM_FALLBACK_TIMEOUT = 2,   /* a wait inside the launch ran out (host thread stalled > MCALF_STREAM_TIMEOUT) */
    MCALF_STREAM_FALLBACK_STARVED = 3    /* the launch drained, but an XCD that was dealt rows had received no workgroup:
                                            its rows were never evaluated, the launch's results were discarded */
};
int mcalf_last_launch(const mcalf_ctx* ctx, mcalf_launch_info_t* info);
/* The same for device entry k of a multi-device context (mcalf_last_launch itself reports entry 0 and devices_used). */
int mcalf_last_launch_sub(const mcalf_ctx* ctx, int32_t k, mcalf_launch_info_t* info);

/* Restrict EVERY stream the context owns -- the launch stream of the host-pointer entries and its auxiliaries, the resident
 * evaluator's stream (its kernel is stopped first; the next one-theta call starts another), the exchange stream of the
 * library's gather (pending exchanges are waited for) -- (*_device entries run on the CALLER's stream) to the
 * compute units of `mask` -- bit i % 32 of word i / 32 = CU i in the runtime's numbering (hipExtStreamCreateWithCUMask; on a
 * multi-XCD device consecutive bits go round the XCDs: bit i = CU i / 8 of XCD i % 8) -- as an embedding application does to
 * share one GPU between ranks.  nwords = 0 removes the mask.  The context waits for its streams, re-creates them and probes
 * again which XCDs they reach: on anything but all eight XCDs of an unpartitioned MI355X large host-pointer batches take the
 * row-block pipeline instead of the streaming launch (mcalf_launch_info_t.xcd_mask / .stream_fallback say so).  Measured on
 * ROCm 7.2 / MI355X (tools/explore/cu_mask_probe.py): an XCD whose share of the mask is EMPTY runs unrestricted, so a mask
 * slows a stream down but cannot take an XCD away from it -- what does is a partition mode (DPX / QPX / CPX).  A masked
 * stream is a BLOCKING stream in HIP's sense (hipExtStreamCreateWithCUMask has no non-blocking form): it synchronises with
 * work on the legacy default stream, which an unmasked context's streams do not.  Results never depend on the mask. */
int mcalf_set_cu_mask(mcalf_ctx* ctx, const uint32_t* mask, int32_t nwords);

/* How a streaming launch deals the rows of a batch to `nxcd` XCDs (blocks of eight rows, block k -> XCD k % nxcd): for
 * every row r < nrows, owner[r] = its XCD and local[r] = its index in that XCD's queue.  Pure host arithmetic (no
 * device needed) -- the same constexpr functions the kernels use; tests check that it is a bijection for every count. */
int mcalf_stream_partition(int32_t nrows, int32_t nxcd, int32_t* owner, int32_t* local);

/* Measurement aid: between _begin and _end every fused-kernel launch of this context is bracketed by
 * HIP events on the stream it is launched on (at most max_launches of them); _end waits for them and
 * returns the mean duration in milliseconds of mcalf_fused_kernel alone (the small per-sample set-up
 * kernel that precedes it is outside the bracket). */
int mcalf_profile_begin(mcalf_ctx* ctx, int32_t max_launches);
int mcalf_profile_end(mcalf_ctx* ctx, double* mean_ms, int32_t* launches);

/* Prior transform: theta = cube * ptp(bounds) + min(bounds) per dimension; when
 * int_ncomp != 0 the ncomp slot is truncated like Python int() (_scale_cube_pc), otherwise
 * left as is (_scale_cube_mn).  lo/hi are [ndim] host arrays; cube/theta [batch][ndim] host. */
int mcalf_scale_cube_batch(mcalf_ctx* ctx, const double* lo, const double* hi, const double* cube,
                           int64_t batch, int32_t int_ncomp, double* theta);

/* Unit cube in -> logL out: the composition lnlhood_pc(_scale_cube_pc(cube))
 * (hires_fitter.py:202-209 + :250-262) without materialising theta on the host.  mcalf_set_prior stores
 * the box (lo/hi [ndim], host, copied) and whether the ncomp slot is truncated (_scale_cube_pc) or not
 * (_scale_cube_mn); the prior transform is then applied inside the per-sample set-up kernel with numpy's
 * rounding (separate multiply and add), so logL is bit-identical to mcalf_loglike_batch on the rows
 * mcalf_scale_cube_batch returns.  theta / dtheta may be NULL; otherwise they receive the transformed
 * rows [batch][ndim] (what the sampler stores as the live point). */
int mcalf_set_prior(mcalf_ctx* ctx, const double* lo, const double* hi, int32_t int_ncomp);
int mcalf_loglike_cube_batch(mcalf_ctx* ctx, const double* cube, int64_t batch, double* theta, double* logL);
int mcalf_loglike_cube_batch_device(mcalf_ctx* ctx, const double* dcube, int64_t batch, double* dtheta,
                                    double* dlogL, void* stream);

/* Multi-GPU inside the library (the reference's only parallelism is data parallelism over live
 * points: PolyChord's MPI workers cli.py:110, jaxns' vmap cli.py:275-280).  One process per GPU, each with its own
 * context; the context owns an RCCL communicator (xGMI on one node).  Rank 0 obtains a 128-byte id with
 * mcalf_comm_unique_id and hands it to the other ranks by any means (MPI, torch.distributed, a file); every rank
 * then calls mcalf_comm_init (collective).
 *
 * mcalf_loglike_gatherv_device evaluates the rank's own contiguous block of live points on `stream` and enqueues ONE
 * grouped send / receive exchange (what ncclGather is) that lands every rank's logL block in dlogL_all on `root`:
 * block r at offset counts[0] + ... + counts[r-1].  `counts` is a HOST array [nranks] that every rank passes
 * identically (ragged shards allowed, zeros allowed; counts[rank] must equal batch_local); NULL means every rank
 * evaluates batch_local rows (mcalf_loglike_gather_device is that form).  dlogL_all may be NULL on the other ranks.
 * By default the exchange is enqueued on `stream` itself, right behind the kernels (plain stream semantics: what
 * follows on `stream` sees the gathered vector; nothing crosses streams).  With mcalf_comm_set_overlap(ctx, 1) and more
 * than one rank it runs on a context-owned stream behind an event of `stream`, so that the root's NEXT kernels do not
 * queue behind receives that wait for its peers: the exchange of call k overlaps the kernels of call k+1, the caller
 * alternates between TWO (dlogL_local, dlogL_all) buffer pairs -- the library orders call k+2 behind exchange k -- and
 * calls mcalf_comm_join(ctx, stream) before it consumes a result.
 * Nothing synchronises the host.  Because a live point's arithmetic does not depend on the shard, the gathered
 * vector equals the single-GPU result bit for bit.
 *
 * Errors never leave a peer waiting:
 *   - MCALF_ERR_INVALID from the argument checks: nothing was enqueued on this rank (a programming error that every
 *     rank of a correct SPMD caller makes alike).
 *   - a LOCAL failure (workspace growth, kernel launch): the rank still takes its part in the exchange with a block
 *     of NaNs, then returns its error code; the peers complete, the root sees NaN rows for that rank, and the
 *     communicator stays usable.
 *   - a failure inside the exchange itself (RCCL / event calls): the communicator is aborted (ncclCommAbort),
 *     MCALF_ERR_COMM is returned and every later gather is refused until mcalf_comm_destroy + mcalf_comm_init.
 * RCCL is loaded at run time by the first of these calls (MCALF_ERR_COMM if it is absent; MCALF_RCCL_LIB names an
 * explicit library); single-GPU use never touches it. */
#define MCALF_COMM_ID_BYTES 128
int mcalf_comm_unique_id(void* id128);
int mcalf_comm_init(mcalf_ctx* ctx, const void* id128, int32_t nranks, int32_t rank);
int mcalf_comm_info(const mcalf_ctx* ctx, int32_t* nranks, int32_t* rank);
int mcalf_comm_destroy(mcalf_ctx* ctx);
int mcalf_comm_set_overlap(mcalf_ctx* ctx, int32_t on);
int mcalf_comm_join(mcalf_ctx* ctx, void* stream);
int mcalf_loglike_gather_device(mcalf_ctx* ctx, const double* dP, int64_t batch_local, double* dlogL_local,
                                double* dlogL_all, int32_t root, void* stream);
int mcalf_loglike_gatherv_device(mcalf_ctx* ctx, const double* dP, int64_t batch_local, double* dlogL_local,
                                 double* dlogL_all, const int64_t* counts, int32_t root, void* stream);

/* Diagnostic: out[i] = H(x[i], y[i]) = Re w(x + i y) evaluated by the device Voigt function
 * (host pointers).  device = -1 for the current device. */
int mcalf_voigt_hjerting(const double* x, const double* y, int64_t n, double* out, int32_t device);
/* Same with the arithmetic an interpolation NODE gets: between x_c (6.2 for K = 1) and 8 the zone-1 wing
 * polynomial with exp(-x^2) dropped, instead of the core table a directly evaluated pixel uses there. */
int mcalf_voigt_hjerting_nodes(const double* x, const double* y, int64_t n, double* out, int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* MCALF_HIP_H */
