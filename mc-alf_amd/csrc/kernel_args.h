// What the device code (kernels.hip) and the host side of libmcalf_hip.so (host_abi.cpp, host_stream.cpp, broker.cpp,
// comm.cpp) share: the launch geometry, the kernel-argument block, the words a streaming or resident launch exchanges
// with the host, and the table of kernel entry points the host launches through hipLaunchKernel.  Plain C++ -- the
// host files are compiled without the HIP language mode.  Part of the kernel-source hash (mc-alf_amd/build.py): a
// change of a layout here changes what the device code does.
#pragma once
#include <cstddef>
#include <cstdint>

#include "voigt_tables.h"

namespace mcalf {

constexpr int kBlock = 512;
constexpr int kPpt = 8;                 // pixels per thread: (tile + halo) <= kBlock * kPpt
constexpr int kExtMax = kBlock * kPpt;  // 4096 pixels = 32 KiB of LDS
constexpr int kWaves = kBlock / 64;
constexpr int kRecStride = 8;           // doubles per (component,line) record in LDS
constexpr int kTabPad = VT_NTOT + 7;    // folded table in LDS: zone0 / zoneF shifted to stay 16-B aligned, then uthr, A, B
constexpr int kZ0Lds = VT_Z0_OFF + 1;
constexpr int kZFLds = VT_ZF_OFF + 2;
constexpr int kLineLds = kZFLds + VT_FDEG + 1;   // [uthr, A, B] of the line, right behind its zone-F coefficients: the
                                                 // node pass reads all ten doubles off ONE base address (5 x ds_read_b128)
static_assert(kLineLds + 3 <= kTabPad && (kLineLds % 2) == 1, "line constants pair up with the last zone-F coefficient");
// Lines folded per workgroup barrier of the component loop: 4 or 5, chosen per context (whichever needs fewer
// barriers for the context's largest line count; measured on MI355X: config C, 20-24 lines, -0.8 % with 5;
// config E, 16 lines, +1.3 % with 5).  The fused kernel is instantiated for both (fused_kernel_ptr).
static_assert(kBlock == 64 * VT_INODES, "one interpolation weight per thread");
static_assert(kPpt == 8, "the skip tests of eval_line treat the eight segments of a wave as two halves");
static_assert(VT_NTOT <= kBlock - 64 && (kZ0Lds % 2) == 0 && (kZFLds % 2) == 0 && (kTabPad % 2) == 0, "LDS table layout");
constexpr int kRedDoubles = 3 * kWaves + 2;   // per-wave partials (sum, count, scratch) + the next work-item index
constexpr int kTileSlack = 16;           // zero-filled entries past the halo (sliding-window over-read)
constexpr size_t kLdsBudget = 78 * 1024;   // two workgroups per CU (160 KiB); needs the MaxDynamicSharedMemorySize attribute
constexpr bool kFarInterp = true;          // far wings at 8 nodes per 64-pixel segment (tools/interp_check.py builds the other variant)
constexpr double kInterpC = 1.0e-3;        // interpolation error <= kInterpC (du/u0)^8 (measured 4.4e-4, tools/ + DESIGN.md)
constexpr double kInterpTol = 1.0e-15;     // allowed optical-depth error per (line, pixel) from the interpolation
constexpr double kCcgs = 2.9979245e10;  // hires_fitter.py:66
constexpr double kFwhmToSigma = 2.354820;   // hires_fitter.py:454
constexpr double kKernelReach = 3.0348;     // hires_fitter.py:458
constexpr double kTauConst = 0.014971475;   // hires_fitter.py:364

enum Mode : int { kModeLogL = 0, kModeModel = 1, kModeChi2 = 2, kModeOneComp = 3 };

struct LineDev {
    double wrest_cm;  // wrest/1e8                 hires_fitter.py:376
    double f;
    double gamma4pi;  // gamma / (4 pi)            hires_fitter.py:361 (a = gamma4pi / dnu)
    double nujk;      // ccgs / wrest_cm           hires_fitter.py:359
};

// Per-sample set-up, one wave per live point: decode, (component,line) records, LSF taps.  Runs once per
// sample ahead of the fused kernel (whose workgroups -- several per sample when the spectrum is tiled --
// then start with plain coalesced loads instead of a chain of dependent loads and libm calls).
struct SampleHdr {
    double cont;     // continuum
    double bot;      // tap sum astropy's loop divides by (1 on the JAX path)
    int ncl;         // records in use
    int n;           // LSF half-width of this sample
    int bad;         // LSF wider than the provisioned halo
    int ngeneral;    // records that need the general Voigt path
};

struct KArgs {
    const double* nu;       // [npix] ccgs / (wl/1e8): pixel frequency at z = 0
    const double* obj;      // [npix]
    const double* ispec2;   // [npix] 1/err^2
    const double* lgis;     // [npix] log(ispec2)
    const double* err;      // [npix] obj_noise (asymmetric veto only)
    const double* P;        // [batch][ndim]  (mode OneComp: [batch][5])
    double* partial;        // [batch][ntiles][4]  (sum, nonzero-count, #resid>4, #resid>5)
    double* out;            // logL / chi2 [batch]   (written directly when ntiles == 1)
    double* model;          // [batch][npix] or nullptr
    const LineDev* lines;   // [nlines] then the filler line at [nlines]
    const double* tabs;     // T[VT_NY][VT_NTOT]
    double* recs;           // [batch][ncl_cap][8] records written by the sample kernel
    double* taps;           // [batch][2 n_cap + 8] normalised LSF taps, zero padded
    SampleHdr* hdr;         // [batch]
    const double* wtab;     // [64][8] Lagrange weights of the far-wing interpolation
    const unsigned long long* segok;   // [ntiles] bit m: 64-pixel segment m of the tile may be interpolated
    int npix, ndim, ntiles, tile, n_cap, ncl_cap;
    int nlines, ncompmax, nfill, startind, endind, freespecres, freecont;
    int targonly, mode, jax_half, onecomp_fill, asymm;
    int taps_shared;        // 1: fixed resolution -> every live point has the same LSF taps, stored once (row 0)
    int selfhalo;           // 1: single-tile spectrum whose halo entries are copies of the tile's own pixels (see fused kernel)
    double specres_fixed, contval_fixed, velstep, log2pi;
    double dnu_seg;         // largest |nu(first) - nu(last)| over the 64-pixel segments
    double veto4, veto5;    // asymmetric veto: allowed counts of resid > 4 / > 5 (threshold + grace)
    // unit-cube input (mcalf_loglike_cube_batch*): P holds cube rows and the prior transform of
    // hires_fitter.py:202-216 is applied while decoding; nullptr = P holds theta
    const double* prior_lo;
    const double* prior_hi;
    double* theta_out;      // [batch][ndim] transformed parameters, or nullptr
    int prior_int;          // 1: int() on the ncomp slot (_scale_cube_pc), 0: leave (_scale_cube_mn)
    // persistent fused kernel: work items (live point x tile) of this launch and its item queue
    int nitems, persist;
    unsigned int* queue;    // reset to 0 by the set-up kernel of the same launch
    // Hand-out order of the persistent kernel (single-tile spectra): ticket t of the queue is live point
    // order[t] -- the live points sorted by their component count, longest first (written by one extra workgroup
    // of the set-up kernel); nullptr = ticket order.  Scheduling only: a live point's arithmetic does not depend on who evaluates it when.
    int* order;
    // Streaming single launch (mcalf_fused_kernel<..., kStream = true>, the host-pointer entries): there is no set-up
    // kernel and no copy command.  The grid sets the live points up itself -- the first `stream_wgs` workgroups keep
    // doing so, row after row as the rows arrive, until none is left, and only then join the others at the item queue
    // -- and a work item is handed to the component loop once its row's stamp says it is set up.
    struct StreamCtl* sctl; // per-XCD item queues and row queues, exit count (self-resetting: the last workgroup out zeroes them)
    unsigned int* ready;    // [batch] ready[s] == gen: live point s is set up (records, taps, header in HBM)
    const unsigned int* arrived;   // rows of P the host has staged so far (page-locked, device-mapped word); nullptr: all
    unsigned int* status;   // page-locked, device-mapped: [0] != 0: a wait ran out (the call fails over), [1] = gen when the grid
                            // has drained, [2] / [3] = fewest / most workgroups an XCD received
    unsigned int gen;       // stamp of this call
    int nrows;              // live points of this launch
    int stream_wgs;         // workgroups PER XCD (the first ones to start there) dedicated to the set-up until the XCD's rows are done
    int eager_rows;         // local BLOCKS (of 8 rows) per XCD that whichever workgroup of the XCD gets there first sets up (the
                            // rows the XCD's first items need; all of them when P is resident in HBM)
    long long spin_ticks;   // longest wait, in ticks of s_memrealtime (100 MHz)
    double* Pdev;           // [batch][ndim] in HBM: where the workgroups copy rows that live in host memory; nullptr: P is in HBM
    int rest_chunk;         // rows a dedicated workgroup claims at a time (a multiple of 8: whole blocks)
    int rec_stride, tap_stride, hdr_stride;   // doubles between the records / taps / headers of consecutive live points in the
                            // streaming workspaces: multiples of a 128-byte line, so that no two live points share one
};

// XCDs of an MI355X in SPX mode.  The streaming launch keeps every hand-over inside one of them: block k of eight rows
// belongs to XCD k % 8, and only workgroups that run there set it up and evaluate it.  On any other device shape -- a DPX /
// QPX / CPX partition, a CU-masked stream -- some of the eight would receive no workgroup, so the host takes the launch
// only where the context's probe kernel saw workgroups on exactly these eight (host_stream.cpp) and otherwise runs the
// row-block pipeline; after every launch it also checks that each XCD did receive workgroups (status[2]).
constexpr int kXcds = 8;
struct StreamCtl {                      // per XCD x: its own queues over ITS live points (blocks of 8 rows, block k -> XCD k % 8)
    unsigned int arrive[kXcds];         // workgroups of the launch that started on XCD x (the first few are its set-up workgroups)
    unsigned int sq_eager[kXcds];       // local blocks claimed of [0, eager_blocks): any workgroup of the XCD
    unsigned int sq_rest[kXcds];        // local blocks claimed of the rest: the XCD's dedicated workgroups
    unsigned int queue[kXcds];          // local tickets handed out by the XCD's item queue
    unsigned int exited;                // workgroups that have left the kernel
    unsigned int fin_exited;            // workgroups of the finalize kernel behind it (tiled spectra) that have finished
};
// status[0] of a streaming launch (0 = nobody gave up)
constexpr unsigned kStreamHostLate = 1u;     // a wait for the host's row count ran out
constexpr unsigned kStreamStampLate = 2u;    // a wait for a row's stamp ran out

// Live points of XCD x of n when the rows are dealt out in blocks of eight (block k -> XCD k % n), and the row
// behind the XCD's local index j.  (constexpr: host and device; tests/test_host_logic.py checks the partition through
// mcalf_stream_partition.)
constexpr int stream_rows_of(int nrows, int x, int n) {
    const int nblocks = (nrows + 7) >> 3;
    if (nblocks <= x) return 0;
    const int nbx = (nblocks - x + n - 1) / n;
    return 8 * nbx - ((nblocks - 1) % n == x ? 8 * nblocks - nrows : 0);
}
constexpr int stream_row(int x, int j, int n) { return 8 * n * (j >> 3) + 8 * x + (j & 7); }

// LDS flux tile, "mod-8 planar": element i lives in plane (i & 7) at index (i >> 3).  The convolution
// thread that owns outputs 8g..8g+7 then reads every plane at consecutive indices with compile-time
// offsets (no address arithmetic, lanes hit consecutive slots); the plane stride 516 == 4 (mod 32)
// also keeps the lane-contiguous flux stores conflict free.
constexpr int kPlaneStride = (kExtMax + kTileSlack) / 8 + 2;      // 516
static_assert(kPlaneStride % 32 == 4, "plane stride must be 4 mod 32");
constexpr int tile_doubles(int) {
    return 8 * kPlaneStride > VT_NY * VT_NTOT ? 8 * kPlaneStride : VT_NY * VT_NTOT;   // the region doubles as the T table
}

constexpr int kSetupBlockMax = 512;
constexpr int kMinWaves = 4;            // waves per SIMD the fused kernel is compiled for (2 workgroups of 8 waves per CU)

// What a result slot of a page-locked block holds until its kernel has written it: a quiet NaN with a payload no
// arithmetic produces (the kernels' NaNs are the canonical one or carry an operand's payload).
constexpr uint64_t kResultPending = 0x7FF8C0DEC0DE0001ull;

// RESIDENT one-theta evaluator (mcalf_resident_kernel): the mailbox it serves and the words the workgroups of one
// resident launch share.
constexpr unsigned kResRunning = 1u, kResLeaving = 2u, kResGone = 3u;
constexpr int kResRowMax = 64;
struct alignas(64) ResidentBox {
    unsigned int req;            // host: number of the request whose row is in `row` (written last, release)
    unsigned int quit;           // host: non-zero = leave at the next look
    unsigned int ack;            // device: number of the last request answered
    unsigned int state;          // device: kResRunning / kResLeaving / kResGone
    double result;               // device; the host fills it with kResultPending before it posts a request
    double pad[5];
    double row[kResRowMax];      // host: the parameter row
};
// Words the workgroups of ONE resident launch share (device memory, zeroed by the host before the launch): the time of
// the launch's last answered request, and the word that tells everybody to leave.  A launch may hold one workgroup (a
// context's own evaluator) or one per mailbox (the broker: workgroup k serves mailbox k); its workgroups leave TOGETHER --
// when workgroup 0 finds that none of them has answered anything for `idle_ticks`, or when a mailbox says `quit` -- so
// that one launch on one stream is all there ever is (hardware queues are few: a launch per mailbox, each on its own
// stream, had the resident kernels of one queue wait for each other's idle limits).
struct ResidentShared {
    unsigned long long last;     // s_memrealtime of the last answer of any workgroup of the launch
    unsigned int leave;          // non-zero: everybody leaves (after a last look at their mailboxes)
    unsigned int pad;
};

// ---- kernel entry points (defined at the end of kernels.hip; the host launches them with hipLaunchKernel) -------------
#define MCALF_INTERNAL __attribute__((visibility("hidden")))      // not part of what the library exports
// The instantiations of the fused kernel: (JAX semantics, self-halo tile, lines per barrier) for batches, the one-launch
// variant (set-up inside the kernel, always 4 lines per barrier) for small calls, the streaming launch of the
// host-pointer entries.
MCALF_INTERNAL const void* fused_kernel_ptr(bool jax, bool selfhalo, int lps, bool inl = false, bool stream = false);
MCALF_INTERNAL int fused_kernel_count();                       // every instantiation, for hipFuncSetAttribute
MCALF_INTERNAL const void* fused_kernel_at(int i);
MCALF_INTERNAL const void* resident_kernel_ptr(bool jax, bool selfhalo);
MCALF_INTERNAL const void* sample_kernel_ptr(bool jax);        // (const KArgs a, long batch)
MCALF_INTERNAL const void* finalize_kernel_ptr();              // (const double* partial, double* out, long batch, int ntiles, int mode, int asymm, double veto4, double veto5, unsigned* fin_count, unsigned* done_word, unsigned gen)
MCALF_INTERNAL const void* hjert_kernel_ptr();                 // (const double* x, const double* y, long n, double* out, const double* tabs, int node_form)
MCALF_INTERNAL const void* scale_cube_kernel_ptr();            // (const double* lo, const double* hi, const double* cube, long total, int ndim, int slot, int int_ncomp, double* theta)
MCALF_INTERNAL const void* xcd_probe_kernel_ptr();             // (unsigned int* mask): ORs 1 << XCC_ID of every workgroup into *mask
// LSF wider than a workgroup tile (kernels.hip, "wide" kernels): taps per live point, convolution + terms from HBM
constexpr int kWideBlockThreads = 256;
constexpr int kWideBlockPix = 8 * kWideBlockThreads;       // output pixels per workgroup of the wide convolution (eight per thread)
constexpr int kWideTapChunk = 512;                         // taps staged in LDS per pass
MCALF_INTERNAL const void* wide_taps_kernel_ptr(bool jax);             // (const KArgs a, double* taps, long tap_stride, SampleHdr* hdr), grid = rows
MCALF_INTERNAL const void* wide_conv_kernel_ptr(bool jax);             // (const KArgs a, const double* flux, const double* taps, long tap_stride, const SampleHdr* hdr, int nblocks), grid = (ceil(npix / kWideBlockPix), rows)
MCALF_INTERNAL const void* wide_rows_kernel_ptr();             // (const double* rows, double* out, long batch): (R, cont, N, z, b) rows with cont := 1

}  // namespace mcalf
