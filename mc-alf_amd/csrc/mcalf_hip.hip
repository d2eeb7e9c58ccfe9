// libmcalf_hip.so -- MI355X (gfx950) implementation of the MC-ALF likelihood hot path.
//
// Per call, on the caller's stream:
//   mcalf_sample_kernel  one wave per live point s: decode p_s (optionally from a unit-cube row), the
//                        (component,line) records, the LSF taps                hires_fitter.py:412-431,357-364,454-459
//   mcalf_fused_kernel   one workgroup per (live point s, pixel tile T):
//     1. tau(pixel) = sum_cl K_cl H(u_cl(pixel), a_cl): per line a node pass (far wings at 8 nodes per 64-pixel
//        segment, interpolated once per sample) and per-pixel evaluation of the rest; flux = exp(-tau) into an
//        LDS tile with +-n halo                                               hires_fitter.py:365,377,430-442
//     2. sliding-window Gaussian LSF from LDS (periodic / zero-pad)           hires_fitter.py:452-464 / :667-681
//     3. x continuum, Gaussian log-likelihood terms, nansum, wave + LDS reduce   hires_fitter.py:292-294
//   mcalf_finalize_kernel  only when a spectrum needs several tiles: adds the per-tile partials in fixed order
//                        (no float atomics anywhere, so a sharded batch equals the unsharded one bit for bit).
//
// The C ABI is declared in include/mcalf_hip.h.  There is no CPU fallback: every entry point
// fails with MCALF_ERR_NODEVICE when no gfx950 device is present.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the library is resolved at run time (mcalf_comm_*), never linked

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mcalf_hip.h"
#include "voigt_device.h"

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
    unsigned int* status;   // page-locked, device-mapped: [0] != 0: a wait ran out (the call fails), [1] = gen when the grid has drained
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

constexpr int kXcds = 8;                // XCDs of an MI355X (the streaming launch keeps every hand-over inside one of them)
struct StreamCtl {                      // per XCD x: its own queues over ITS live points (blocks of 8 rows, block k -> XCD k % 8)
    unsigned int arrive[kXcds];         // workgroups of the launch that started on XCD x (the first few are its set-up workgroups)
    unsigned int sq_eager[kXcds];       // local blocks claimed of [0, eager_blocks): any workgroup of the XCD
    unsigned int sq_rest[kXcds];        // local blocks claimed of the rest: the XCD's dedicated workgroups
    unsigned int queue[kXcds];          // local tickets handed out by the XCD's item queue
    unsigned int exited;                // workgroups that have left the kernel
};

// The XCD a wave runs on (XCC_ID, bits 3:0 of hardware register 20 on gfx942 / gfx950).
__device__ __forceinline__ int xcd_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (kXcds - 1)); }
// Live points of XCD x when the rows are dealt out in blocks of eight (block k -> XCD k % kXcds), and the row behind the
// XCD's local index j.
__device__ __forceinline__ int stream_rows_of(int nrows, int x) {
    const int nblocks = (nrows + 7) >> 3;
    if (nblocks <= x) return 0;
    const int nbx = (nblocks - x + kXcds - 1) / kXcds;
    return 8 * nbx - ((nblocks - 1) % kXcds == x ? 8 * nblocks - nrows : 0);
}
__device__ __forceinline__ int stream_row(int x, int j) { return 64 * (j >> 3) + 8 * x + (j & 7); }

// LDS flux tile, "mod-8 planar": element i lives in plane (i & 7) at index (i >> 3).  The convolution
// thread that owns outputs 8g..8g+7 then reads every plane at consecutive indices with compile-time
// offsets (no address arithmetic, lanes hit consecutive slots); the plane stride 516 == 4 (mod 32)
// also keeps the lane-contiguous flux stores conflict free.
constexpr int kPlaneStride = (kExtMax + kTileSlack) / 8 + 2;      // 516
static_assert(kPlaneStride % 32 == 4, "plane stride must be 4 mod 32");
__device__ __forceinline__ int tile_pos(int i) { return (i & 7) * kPlaneStride + (i >> 3); }
__host__ __device__ constexpr int tile_doubles(int) {
    return 8 * kPlaneStride > VT_NY * VT_NTOT ? 8 * kPlaneStride : VT_NY * VT_NTOT;   // the region doubles as the T table
}

// Value of lane (l - N) within each row of 16 lanes (0 where there is none): one v_mov_b32_dpp per half,
// no LDS round trip (ds_bpermute, which __shfl_* compiles to, costs an LDS latency per step).
template <int kCtrl, int kRowMask = 0xF>
__device__ __forceinline__ double dpp_move(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, kCtrl, kRowMask, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), kCtrl, kRowMask, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// Sum over the 64 lanes of a wave; the total ends up in lane 63 (DPP row shifts + row broadcasts).
__device__ __forceinline__ double wave_sum_to_last(double v) {
    v += dpp_move<0x111>(v);          // row_shr:1
    v += dpp_move<0x112>(v);          // row_shr:2
    v += dpp_move<0x114>(v);          // row_shr:4
    v += dpp_move<0x118>(v);          // row_shr:8   -> lane 15 of every row holds the row sum
    v += dpp_move<0x142, 0xA>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xC>(v);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Tell the compiler a 64-bit value is wave-uniform (keeps it in SGPRs, branches on it are scalar).
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Sum over the 64 lanes of a wave, result in every lane (as a wave-uniform value): the DPP reduction above and
// one v_readlane per half -- no LDS round trips (a __shfl_xor butterfly is 12 ds_bpermute, each an LDS latency
// on the serial path of the set-up kernel).
__device__ __forceinline__ double wave_allsum(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, wave_sum_to_last(v));
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, 63), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Sum over the workgroup, result in every thread; fixed order (deterministic).
__device__ __forceinline__ double block_sum(double v, double* scratch, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += scratch[w];
    __syncthreads();
    return s;
}

__device__ __forceinline__ double finalize_value(int mode, double sum, double nnz, bool asymm, double c4,
                                                 double c5, double veto4, double veto5) {
    if (mode == kModeChi2) return (nnz == 0.0) ? INFINITY : sum;   // hires_fitter.py:241-246
    if (asymm && (c5 > veto5 || c4 > veto4)) return -INFINITY;     // :296-303
    return -0.5 * sum;                                             // :294
}

// 10^x as exp(x ln 10) with the product carried in two doubles (about 1 ulp, a fraction of the cost of pow()).
__device__ __noinline__ double pow10_edge(double x) { return pow(10.0, x); }   // +-inf, NaN, over/underflow (10**-inf = 0)
__device__ __forceinline__ double pow10_fast(double x) {
    constexpr double kLn10Hi = 2.302585092994045901, kLn10Lo = -2.1707562233822494e-16;
    if (!(fabs(x) <= 300.0)) return pow10_edge(x);       // the library's edge cases, kept out of line
    const double p = x * kLn10Hi;
    const double e = fma(x, kLn10Hi, -p) + x * kLn10Lo;
    const double r = exp_neg(-p);                        // == exp(p) to the last bit (voigt_device.h), a third of the code
    return fma(r, e, r);
}

// Record per (component,line): [A, B, x2c, y, K, Kyt, Kgen, uthr]   (general-path lines: [A, B, y, 0, 0, 0, K, 0])
//   u = nu*A - B;  tau += K H(u, y);  Kyt = K y / sqrt(pi) scales the wing polynomials;
//   x2c = x_c^2: beyond it exp(-x^2) is below 2e-17 in optical depth, so the line is pure wing there (the
//        interpolation threshold never lies inside it; nodes between x_c and 8 use the zone-1 polynomial);
//   Kgen != 0 -> general path (eval_general_lines);
//   uthr: a 64-pixel segment whose pixels all have |u| >= uthr is evaluated at 8 nodes and interpolated.
__device__ inline void build_line_record(double* rec, double logN, double z, double b_kms, const LineDev& ln,
                                         double dnu_seg) {
    const double cold = pow10_fast(logN);                // :357  10.0**N
    const double zp1 = z + 1.0;                          // :358
    // ONE division per record: 1/dnu = wrest / b (:360 with :376's b*1e5); everything that the reference divides
    // by dnu is a multiple of it (five IEEE divisions were a third of this kernel's serial path; the products
    // differ from the quotients by an ulp, far below the 1e-11 cancellation noise u carries anyway)
    const double rdnu = ln.wrest_cm / (b_kms * 1e5);
    const double a = ln.gamma4pi * rdnu;                 // :361  gamma / (4 pi dnu)
    const double cne = kTauConst * cold * ln.f;          // :364
    const double K = cne * rdnu;                         // :365  tau = cne * H / dnu
    rec[0] = zp1 * rdnu;                                 // u = ((c/(lam/zp1)) - nujk)/dnu  (:362)
    rec[1] = ln.nujk * rdnu;
    rec[2] = core_limit_x2(K);
    rec[3] = a;
    rec[4] = K;
    rec[5] = K * a * kInvSqrtPi;
    double flag = 0.0;
    if (!(a <= kYFastMax) || !(a >= 0.0)) flag = 1.0;    // general path (also NaN)
    else if (K * 1.6e-28 > 2e-17) flag = 1.0;            // absurd columns: exp(-x^2) matters past |x| = 8
    rec[6] = 0.0;
    // interpolation error kInterpC (du/u0)^8 Kyt/u0^2 <= kInterpTol  ->  u0^10 >= kInterpC Kyt du^8 / tol
    const float du = (float)(rec[0] * dnu_seg);
    const float du2 = du * du, du4 = du2 * du2;
    const float q = (float)(kInterpC / kInterpTol) * (float)rec[5] * du4 * du4;
    // q^0.1 as exp2(0.1 log2 q) on the hardware's v_log_f32 / v_exp_f32 (q >= 1, so no denormal case; their ~1e-7
    // relative error is nothing against the margin) -- powf() expands to ~150 instructions of this kernel's serial path
    double uthr = (double)__builtin_amdgcn_exp2f(0.1f * __builtin_amdgcn_logf(fmaxf(q, 1.0f))) * 1.02;   // 2 % margin over the float estimate
    uthr = fmax(uthr, (double)__fsqrt_rn((float)rec[2]) * 1.000001);           // never inside the core table's range (x2c in [36, 64])
    rec[7] = !(uthr < 1e30) ? INFINITY : uthr;
    if (flag != 0.0) {
        // General-path line: the hot loop carries no test for it.  Its fast-path view is a line of zero
        // strength (folded tables all zero, every segment "interpolated"), and eval_general_lines() finds
        // the real damping parameter in slot 2 and the real K in slot 6 (K != 0 marks the record).
        rec[2] = a; rec[3] = 0.0; rec[4] = 0.0; rec[5] = 0.0; rec[6] = K; rec[7] = 0.0;
    }
    // |1+z| beyond 1e100 (or infinite): every |u| overflows, the reference's wofz returns 0 and the line adds
    // nothing (tau < 1e-200).  Written out as a record that contributes exact zeros, because 1/u^2 -> 0 would
    // put 0 * inf = NaN through the reciprocal's Newton step.  NaN parameters still propagate as NaN.
    if (!(fabs(zp1) < 1e100) && zp1 == zp1 && fabs(K) < 1e100 && fabs(a) < 1e100) {
        rec[0] = 0.0; rec[1] = -1e6; rec[2] = 36.0; rec[3] = 0.0; rec[4] = 0.0; rec[5] = 0.0; rec[6] = 0.0;
        rec[7] = 0.0;
    }
}

// acc += a * b and acc += a with the accumulator tied to its register: without the tie the compiler
// gives every update of the thread's 8 running optical depths a fresh register and copies all of them
// back at the loop back-edge (16 v_mov_b64 per line).
__device__ __forceinline__ void fmac_inplace(double& acc, double a, double b) {
    asm("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void add_inplace(double& acc, double a) {
    asm("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(a));
}

// tau[j] += K H(u_j, y) for the thread's kPpt pixels and one (component,line); `tab` is the line's
// folded table in LDS (coefficients, then the line's threshold and the (A, B) of u = nu A - B).
__device__ __forceinline__ void eval_line(const double* __restrict__ tab,
                                          const double (&nu)[kPpt], double (&tau)[kPpt], double nuNode,
                                          double& farNode, unsigned long long segOk) {
    const double A = tab[kLineLds + 1], B = tab[kLineLds + 2];
    double cF[VT_FDEG + 1];
#pragma unroll
    for (int k = 0; k <= VT_FDEG; ++k) cF[k] = tab[kZFLds + k];
    // Node pass.  Lane l holds node (l & 7) of the wave's segment (l >> 3).  A segment whose eight
    // nodes (both end pixels included, u monotonic along it) all have u >= uthr, or all u <= -uthr,
    // is wing (|u| >= x_c: no exp(-u^2)) for this line everywhere: its contribution is evaluated at the nodes only and
    // interpolated to the 64 pixels once, after the component loop.  `done` has bit 8 j set when
    // segment j was handled that way.
    unsigned long long done = 0;
    if (kFarInterp) {
        const double uthr = tab[kLineLds];
        const double un = fma(nuNode, A, -B);
        // issued ahead of the scalar mask chain below, which then runs in the shadow of the reciprocal
        const double x2n = un * un;
        // lanes outside `mine` only need to stay finite (mine lanes have x2n >= uthr^2 >= 36; 4.0 is an inline
        // constant, 36.0 costs two scalar moves per line)
        const double t = fast_rcp(fmax(x2n, 4.0));
        unsigned long long mp = __builtin_amdgcn_ballot_w64(un >= uthr);
        unsigned long long mn = __builtin_amdgcn_ballot_w64(un <= -uthr);
        // u is monotonic along a segment (segOk excludes the wrapped ones), so its two END nodes -- lanes 8j
        // and 8j+7, the segment's first and last pixel -- decide for all eight.
        mp &= mp >> 7;                                        // bit 8j = first and last node
        mn &= mn >> 7;
        done = uniform64((mp | mn) & segOk);                 // (segOk carries bits 8j only, so `done` does too)
        if (done != 0) {                                      // wave-uniform
            // byte j -> 0xFF: one bit per lane.  On 32-bit halves (no carry can cross: 0x01010101 * 0xFF = 0xFFFFFFFF),
            // which is two scalar multiplies instead of a 64-bit one.
            const unsigned long long lanes = ((unsigned long long)((unsigned)(done >> 32) * 0xFFu) << 32) | ((unsigned)done * 0xFFu);
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(lanes);   // the scalar mask IS the lane predicate
            double P;
            if (!mine || x2n >= kX2Far) {                         // (lanes of directly evaluated segments never need a wing zone)
                P = cF[VT_FDEG];
#pragma unroll
                for (int k = VT_FDEG - 1; k >= 0; --k) P = fma(P, t, cF[k]);
            } else {
                const bool z0 = x2n >= kX2Wing;
                const double sv = z0 ? t : fma(t, VT_Z1_A, VT_Z1_B);
                const double* cw = tab + (z0 ? kZ0Lds : VT_Z1_OFF);
                P = cw[VT_WDEG];
#pragma unroll
                for (int k = VT_WDEG - 1; k >= 0; --k) P = fma(P, sv, cw[k]);
            }
            fmac_inplace(farNode, mine ? t : 0.0, P);
        }
    }
    // Tested on 32-bit halves (one s_bitcmp1_b32 + branch per segment), four segments at a time first: a line's
    // core covers one or two ADJACENT segments of a wave, so one half of the eight is usually interpolated throughout.
    const unsigned doneLo = (unsigned)done, doneHi = (unsigned)(done >> 32);
#pragma unroll
    for (int h = 0; h < kPpt / 4; ++h) {
    const unsigned dh = h ? doneHi : doneLo;
    if (dh == 0x01010101u) continue;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;
        if (__builtin_expect(((dh >> (8 * jj)) & 1u) != 0u, 1)) continue;   // whole segment interpolated (wave-uniform, the usual case)
        const double u = fma(nu[j], A, -B);
        const double x2 = u * u;
        // Every branch leaves (t, P) with contribution t * P, and the running optical depth is updated at ONE
        // place after the branches merge: an update inside each branch makes the compiler copy tau[j] into a
        // scratch pair and back (3 vector moves per evaluation).
        double t, P;
        if (x2 >= kX2Far) {                           // |u| >= 16
            t = fast_rcp(x2);
            P = cF[VT_FDEG];
#pragma unroll
            for (int k = VT_FDEG - 1; k >= 0; --k) P = fma(P, t, cF[k]);
        } else if (x2 >= kX2Wing) {                   // 8 <= |u| < 16: polynomial in 1/u^2 (broadcast reads)
            t = fast_rcp(x2);
            const double* cw = tab + kZ0Lds;
            P = cw[VT_WDEG];
#pragma unroll
            for (int k = VT_WDEG - 1; k >= 0; --k) P = fma(P, t, cw[k]);
        } else {                                      // |u| < 8: core table (per-lane LDS gather).  It is valid up to 8, so the
                                                      // pixels between x_c and 8 come here too instead of splitting the wave over
                                                      // a third path (zone 1 serves the interpolation nodes only)
            // |u| < 8 here (NaN converts to 0), so the interval index needs no clamp; s = 8|u| - (2j+1) from 4|u|
            // with the inline constant 2.0 (same value bit for bit, 8.0 costs two scalar moves per segment)
            const double x4 = fabs(u) * 4.0;
            const int jx = (int)x4;
            const double sv = fma(x4, 2.0, -(double)(2 * jx + 1));
            const double* cc = tab + jx * VT_CSTRIDE;
            P = cc[VT_CDEG];
#pragma unroll
            for (int k = VT_CDEG - 1; k >= 0; --k) P = fma(P, sv, cc[k]);
            t = 1.0;                                  // tau += 1 * P rounds exactly like tau += P
        }
        fmac_inplace(tau[j], t, P);
    }
    }
}

// theta[i] of one sample: either the row element itself or, with unit-cube input, cube*ptp + min with the
// separately rounded multiply and add numpy performs (hires_fitter.py:206 / :214) and int() on the ncomp slot.
__device__ __forceinline__ double sample_param(const KArgs& a, const double* __restrict__ p, int i) {
    double v = p[i];
    if (a.prior_lo) {
        const double lo = a.prior_lo[i], hi = a.prior_hi[i];
        {
#pragma clang fp contract(off)
            const double scaled = v * (hi - lo);
            v = scaled + lo;
        }
        if (a.prior_int && i == a.startind) v = trunc(v);        // :207-208
    }
    return v;
}

constexpr int kSetupBlockMax = 512;
constexpr int kOrderBuckets = 64;       // component counts 0 .. 62 get a bucket each, larger ones share the last
constexpr int kOrderKeys = 16;          // keys a thread of the ordering workgroup holds at a time

// Active components of live point s: int(p[startind]) on the numpy path (:428), floor on the JAX path (:616),
// clamped to [0, ncompmax].
template <bool kZeroPad>
__device__ __forceinline__ int sample_ncomp(const KArgs& a, long s) {
    const double ncv = sample_param(a, a.P + (size_t)s * a.ndim, a.startind);
    const double nct = kZeroPad ? floor(ncv) : trunc(ncv);
    return (nct >= 1.0) ? ((nct >= (double)a.ncompmax) ? a.ncompmax : (int)nct) : 0;
}

// One workgroup of the set-up kernel: counting sort of the live points by component count, most components
// first, into a.order.  The fused kernel's queue then hands out similar work items next to each other (the two
// workgroups that share a CU run evenly matched items: measured -2.4 % kernel time at config C with rows
// sorted on the host) and the shortest items last.  LDS atomics only; the order inside a bucket is whatever the
// atomics give, which changes who evaluates a live point, never its value.
template <bool kZeroPad>
__device__ void build_order(const KArgs& a, long batch) {
    __shared__ int hist[kOrderBuckets];
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid < kOrderBuckets) hist[tid] = 0;
    __syncthreads();
    const long chunk = (long)kOrderKeys * nthr;
    auto load_keys = [&](long base, int (&key)[kOrderKeys]) {
#pragma unroll
        for (int k = 0; k < kOrderKeys; ++k) {                     // independent loads: one memory round trip per chunk
            const long s = base + (long)k * nthr + tid;
            key[k] = (s < batch) ? min(a.ncompmax - sample_ncomp<kZeroPad>(a, s), kOrderBuckets - 1) : -1;
        }
    };
    int key[kOrderKeys];
    // pass 1: bucket counts
    for (long base = 0; base < batch; base += chunk) {
        load_keys(base, key);
#pragma unroll
        for (int k = 0; k < kOrderKeys; ++k)
            if (key[k] >= 0) atomicAdd(&hist[key[k]], 1);
    }
    __syncthreads();
    // counts -> first position of each bucket: exclusive prefix sum over the 64 buckets by one wave
    if (tid < 64) {
        static_assert(kOrderBuckets == 64, "one bucket per lane");
        const int c = hist[tid];
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (tid >= off) incl += up;
        }
        hist[tid] = incl - c;
    }
    __syncthreads();
    // pass 2: positions.  A batch of one chunk (4096 live points with 256 threads) still holds its keys.
    for (long base = 0; base < batch; base += chunk) {
        if (batch > chunk) load_keys(base, key);
#pragma unroll
        for (int k = 0; k < kOrderKeys; ++k)
            if (key[k] >= 0) a.order[atomicAdd(&hist[key[k]], 1)] = (int)(base + (long)k * nthr + tid);
    }
}

// The set-up of live point s by ONE wave (`lane` = 0..63): decode, records, taps, header, written through the given
// pointers -- the context's workspaces in HBM (mcalf_sample_kernel) or the workgroup's own LDS (the one-launch
// variant of the fused kernel that small calls use).  One body, so both give the same bits.
template <bool kZeroPad>
__device__ __forceinline__ void setup_sample(const KArgs& a, long s, int lane, double* recs, double* taps, bool writeTaps,
                                             SampleHdr* hdrOut, bool writeTheta) {
    const int rowlen = (a.mode == kModeOneComp) ? 5 : a.ndim;
    const double* p = a.P + (size_t)s * rowlen;
    // ---- 1. decode the parameter vector ---------------------------------------------------
    double R, cont;
    int nc, nfill_eff;
    if (a.mode == kModeOneComp) {                       // hires_fitter.py:379-406
        R = p[0];
        cont = p[1];
        nc = 1;
        nfill_eff = 0;
    } else {
        R = a.freespecres ? sample_param(a, p, 0) : a.specres_fixed;     // :412-417
        cont = a.freecont ? sample_param(a, p, a.freespecres ? 1 : 0) : a.contval_fixed;   // :419-425
        const double ncv = sample_param(a, p, a.startind);
        if (a.theta_out && writeTheta)
            for (int i = lane; i < a.ndim; i += 64) a.theta_out[(size_t)s * a.ndim + i] = sample_param(a, p, i);
        // numpy path: int() truncates (:428); JAX path: floor (:616)
        const double nct = kZeroPad ? floor(ncv) : trunc(ncv);
        nc = (nct >= 1.0) ? ((nct >= (double)a.ncompmax) ? a.ncompmax : (int)nct) : 0;
        nfill_eff = a.targonly ? 0 : a.nfill;           // :437
    }
    // onecomp_fill: 0 = every line of the component, 1 = the filler line, 2 + k = line k alone
    const int nl_eff = (a.mode == kModeOneComp && a.onecomp_fill) ? 1 : a.nlines;
    const int ncl = nc * nl_eff + nfill_eff;

    // One slot per POSSIBLE (component, line) and filler, so that every load below is independent of the
    // sample's ncomp (one memory round trip); the record lands at its compacted index afterwards.
    const int nTargetSlots = (a.mode == kModeOneComp) ? nl_eff : a.ncompmax * a.nlines;
    const int nSlots = nTargetSlots + ((a.mode == kModeOneComp) ? 0 : a.nfill);
    int ngenLane = 0;
    for (int slot = lane; slot < nSlots; slot += 64) {
        double logN, z, b;
        const LineDev* ln;
        int dst;                                    // index in the compacted record list, -1: inactive
        if (a.mode == kModeOneComp) {
            logN = p[2]; z = p[3]; b = p[4];
            ln = (a.onecomp_fill == 0) ? (a.lines + slot)
               : (a.onecomp_fill == 1) ? (a.lines + a.nlines) : (a.lines + (a.onecomp_fill - 2));
            dst = slot;
        } else if (slot < nTargetSlots) {
            const int c = slot / a.nlines;
            const int l = slot - c * a.nlines;
            const int q = 1 + 3 * c + a.startind;               // :431  (N, z, b)
            logN = sample_param(a, p, q); z = sample_param(a, p, q + 1); b = sample_param(a, p, q + 2);
            ln = a.lines + l;
            dst = (c < nc) ? slot : -1;                         // components >= int(p[startind]) are skipped (:430)
        } else {
            const int k = slot - nTargetSlots;
            const int q = 3 * k + a.endind;                     // :439
            logN = sample_param(a, p, q); z = sample_param(a, p, q + 1); b = sample_param(a, p, q + 2);
            ln = a.lines + a.nlines;
            dst = (nfill_eff > 0) ? nc * a.nlines + k : -1;
        }
        double rec[kRecStride];
        build_line_record(rec, logN, z, b, *ln, a.dnu_seg);
        ngenLane += __popcll(__ballot(dst >= 0 && rec[6] != 0.0));        // (wave-uniform count)
        if (dst >= 0) {
#pragma unroll
            for (int k = 0; k < kRecStride; ++k) recs[dst * kRecStride + k] = rec[k];
        }
    }

    // ---- LSF taps --------------------------------------------------------------------------
    int n;          // half-width in pixels
    bool bad = false;
    const double sigma = (R / kFwhmToSigma) / a.velstep;        // :454 / :667
    if (kZeroPad) {
        n = a.jax_half;                                         // :549-560 fixed grid
    } else if (R > a.velstep) {                                 // :445
        const double nd = ceil(kKernelReach * sigma);           // :458
        if (!(nd <= (double)a.n_cap)) { bad = true; n = 0; }
        else n = (int)nd;                                       // x_size = int(2n)+1  (:459)
    } else {
        n = 0;
    }
    // Every wave computes the (few) taps itself, so the normalisation needs no workgroup barrier;
    // wave 0 writes them.  astropy normalises the kernel by its sum and its C loop then divides by
    // the tap sum it accumulates next to the data sum (`bot`); the JAX path only normalises (:670).
    const int ntap8 = (2 * n + 1 + 7) & ~7;
    const double inv2s2 = kZeroPad ? 1.0 / (2.0 * sigma * sigma) : 0.5 / (sigma * sigma);
    const double amp = kZeroPad ? 1.0 : 1.0 / (sqrt(2.0 * M_PI) * sigma);          // Gaussian1DKernel amplitude
    double wsum = 0.0, botOrdered = 0.0;
    if (ntap8 <= 64) {                               // the usual case: one tap per lane, one exp
        const double dk = (double)(lane - n);
        const double g = (lane > 2 * n) ? 0.0 : ((n == 0 && !kZeroPad) ? 1.0 : exp_neg((dk * dk) * inv2s2) * amp);
        const double gsum = wave_allsum(g);
        wsum = g / gsum;
        if (lane < ntap8 && writeTaps) taps[lane] = wsum;
        // astropy's loop adds the taps up next to the data sum, tap after tap (`bot`), and divides by that: formed
        // here in the SAME order as the fused kernel's numerator chain (tap 0 first), so that a constant model comes
        // out of the convolution as exactly that constant, as it does in the reference (hires_fitter.py:463-464)
        // (unrolled over the 64 lanes with constant lane numbers, a scalar trip count and an early exit: as a counted loop
        // over the per-lane n it ran under exec masks with a vector compare per step, 2.4 us on the set-up kernel)
        if (!kZeroPad) {
            const unsigned long long wb = __builtin_bit_cast(unsigned long long, wsum);
            const int last = __builtin_amdgcn_readfirstlane(2 * n);      // (the same in every lane: say so, or the loop runs under exec masks)
#pragma unroll
            for (int k = 0; k < 64; ++k) {                   // (no `break`: a constant trip count is what lets it unroll)
                if (k <= last) {
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)wb, k), hi = __builtin_amdgcn_readlane((unsigned)(wb >> 32), k);
                    botOrdered += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                }
            }
        }
    } else {
        double gsum = 0.0;
        for (int k = lane; k <= 2 * n; k += 64) {
            const double dk = (double)(k - n);
            gsum += exp_neg((dk * dk) * inv2s2) * amp;                                  // :669 / Gaussian1D
        }
        gsum = wave_allsum(gsum);
        for (int k = lane; k < ntap8; k += 64) {
            const double dk = (double)(k - n);
            const double w = (k <= 2 * n) ? exp_neg((dk * dk) * inv2s2) * amp / gsum : 0.0;   // zero-padded to 8
            if (writeTaps) taps[k] = w;
        }
        // (more than 64 taps: rare; every lane repeats the tap expression for the tap-ordered sum)
        if (!kZeroPad)
            for (int k = 0; k <= 2 * n; ++k) {
                const double dk = (double)(k - n);
                botOrdered += exp_neg((dk * dk) * inv2s2) * amp / gsum;
            }
    }
    const double bot = kZeroPad ? 1.0 : botOrdered;
    const int ngen = ngenLane;
    if (lane == 0) {
        SampleHdr h;
        h.cont = cont; h.bot = bot; h.ncl = ncl; h.n = n; h.bad = bad ? 1 : 0; h.ngeneral = ngen;
        *hdrOut = h;
    }
}

// ---- streaming single launch: hand-over between waves, waits ---------------------------------------------------------
// Waves of ONE launch hand data to each other here (records / taps / header of a live point, its stamp, the HBM copy of
// its parameter row), and the host hands rows to the launch while it runs.  The eight XCDs of an MI355X each have their
// own L2, which is not coherent with the others' for ordinary device memory: across XCDs a hand-over needs an agent-scope
// release (write back the producer's L2) and acquire (invalidate the consumer's) -- measured here at 4x the launch's
// duration when done per row and per item.  So nothing is handed over ACROSS XCDs: the live points are dealt out to the
// XCDs in blocks of eight rows (block k -> XCD k % 8), every XCD sets up ITS rows with its own workgroups and consumes
// them with its own workgroups through its own queue (xcd_id(): the hardware's XCC_ID, not an assumption about the
// dispatch order).  Producer and consumer of a row share one L2, which IS coherent: plain stores, a wait for their
// acknowledgement (s_waitcnt vmcnt(0): they are in the L2), then the stamp; the consumer sees the stamp and reads the row
// with plain loads.  Rows own their 128-byte lines, so a consumer's L1 never holds a line of a row it has not been handed.
// Stamps and queue counters are device-scope atomics, which meet in memory.  The host's rows and words are page-locked
// coherent memory, read past every cache; RESULTS go to page-locked memory as system-scope stores (write-through), because
// the host reads them as soon as the completion word says so, ahead of the end-of-kernel write-back -- plain stores were
// measured to linger in one XCD's L2 past that word (rows of one XCD missing from the first call's results).
__device__ __forceinline__ void stream_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void stream_compiler_barrier() { asm volatile("" ::: "memory"); }

// Every wait is bounded (a.spin_ticks of the 100 MHz s_memrealtime clock): a wave that runs out of patience raises
// status[0], after which nobody waits any more -- the rows still missing are published as unusable (`bad`: logL = -inf)
// and the grid drains; the host sees status[0] and fails the call.  No wave can stay behind in the kernel.
__device__ __forceinline__ bool stream_gave_up(const KArgs& a) {
    return __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}

// Wait until the host has staged row r (a.arrived counts the rows staged so far).  false: gave up.
__device__ __forceinline__ bool stream_wait_arrived(const KArgs& a, unsigned r, unsigned& seen) {
    if (seen > r) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (true) {
        seen = __hip_atomic_load(a.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        stream_compiler_barrier();                       // (the row's loads are issued behind this one's return)
        if (seen > r) return true;
        if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > a.spin_ticks || stream_gave_up(a)) {
            __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(32);                    // (~1 us: a poll is a PCIe read)
    }
}

// Wait until live point s is set up (every lane of the workgroup calls this with the same s).  Its records, taps
// and header were written by a wave of this XCD, into the L2 both share, before its stamp.
__device__ __forceinline__ void stream_wait_ready(const KArgs& a, int s, unsigned early) {
    if (early != a.gen) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(a.ready + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.gen) {
            if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > a.spin_ticks) {   // (never seen: the producers' own waits are bounded)
                __hip_atomic_store(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    stream_compiler_barrier();
}

template <bool kZeroPad>
__global__ __launch_bounds__(kSetupBlockMax) void mcalf_sample_kernel(const KArgs a, long batch) {
    // one WAVE per live point, blockDim.x / 64 live points per workgroup (the waves never synchronise); with an
    // ordered hand-out workgroup 0 builds the order and the live points start at workgroup 1
    int blk = blockIdx.x;
    if (a.order) {
        if (blk == 0) { build_order<kZeroPad>(a, batch); return; }
        --blk;
    }
    const long s = (long)blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= batch) return;
    const int lane = threadIdx.x & 63;
    if (s == 0 && lane == 0) *a.queue = 0u;          // item queue of the fused kernel that follows on the stream
    setup_sample<kZeroPad>(a, s, lane, a.recs + (size_t)s * a.ncl_cap * kRecStride,
                           a.taps + (a.taps_shared ? 0 : (size_t)s * (2 * a.n_cap + 8)), !a.taps_shared || s == 0, a.hdr + s, true);
}

// Lines outside the fast path's damping range (flag != 0; none for physical resonance lines).  Kept out
// of the hot loop: the call to the non-inlined general Voigt routine would otherwise pin the running
// optical depths in callee-saved registers and cost a register shuffle per line.
__device__ __forceinline__ void eval_general_lines(const double* __restrict__ sRec, int ncl, const double (&nu)[kPpt],
                                                double (&tau)[kPpt]) {
    for (int cl = 0; cl < ncl; ++cl) {
        const double* rec = sRec + cl * kRecStride;
        if (rec[6] == 0.0) continue;
        const double A = rec[0], B = rec[1], y = rec[2], K = rec[6];
#pragma unroll 1
        for (int j = 0; j < kPpt; ++j) {
            const double u = fma(nu[j], A, -B);
            tau[j] = fma(K, hjert_general(fabs(u), y), tau[j]);
        }
    }
}

// Streaming single launch, set-up phase of a workgroup on XCD x (one WAVE per live point, as in mcalf_sample_kernel; same
// setup_sample(), same bits).  A workgroup claims local blocks (eight consecutive rows each) of ITS XCD with one atomic --
// one of the first `eager_rows` blocks, which any workgroup of the XCD may take; then, a dedicated workgroup only (the
// first `stream_wgs` to start on the XCD), `rest_chunk` / 8 of the remaining ones, in ticket order, until none is left.
// Thread 0 alone asks the host's row count (every wave polling a word of host memory saturated the PCIe read queue the
// rows themselves come through).  Rows that live in host memory are first copied to HBM by the whole workgroup --
// coalesced, all loads of the claim in flight at once, ONE PCIe round trip instead of setup_sample's two or three
// dependent ones -- and set up from the copy.  Claims are dynamic on purpose: a row is owned by a workgroup that is
// running, never by one that waits for a slot.
template <bool kZeroPad>
__device__ __forceinline__ void stream_setup_phase(const KArgs& a, int tid, int x, int* sClaim) {
    const int lane = tid & 63, wave = tid >> 6;
    const int nloc = stream_rows_of(a.nrows, x), nblk = (nloc + 7) >> 3;      // this XCD's live points / local blocks
    const int eager = min(a.eager_rows, nblk);
    KArgs as = a;
    if (a.Pdev) as.P = a.Pdev;                           // the rows are set up from their copy in HBM
    unsigned seen = a.arrived ? 0u : (unsigned)a.nrows;  // (thread 0's view of the host's row count)
    if (tid == 0) sClaim[3] = (int)atomicAdd(&a.sctl->arrive[x], 1u);
    __syncthreads();
    const bool dedicated = sClaim[3] < a.stream_wgs;
    bool rest = false;
    // A dedicated workgroup shares its CU with a workgroup that is in the component loop at raised priority; the
    // set-up is a chain of latencies with few instructions: it goes first, and the queue stays ahead of the consumers.
    __builtin_amdgcn_s_setprio(3);
    while (true) {
        if (tid == 0) {
            const int want = rest ? max(a.rest_chunk >> 3, 1) : 1, lim = rest ? nblk : eager;
            const int c = rest ? eager + (int)atomicAdd(&a.sctl->sq_rest[x], (unsigned)want) : (int)atomicAdd(&a.sctl->sq_eager[x], 1u);
            const int cnt = c < lim ? min(want, lim - c) : 0;
            int ok = 1;
            if (cnt != 0 && a.arrived) {                 // (rows arrive in order: the claim's last row is the one to wait for)
                const int last = min(stream_row(x, 8 * (c + cnt) - 1), a.nrows - 1);
                ok = stream_wait_arrived(a, (unsigned)last, seen) ? 1 : 0;
            }
            sClaim[0] = c; sClaim[1] = cnt; sClaim[2] = ok;
        }
        __syncthreads();
        const int c = sClaim[0], cnt = sClaim[1];
        const bool ok = sClaim[2] != 0;
        __syncthreads();                                 // (the slots are rewritten by the next claim)
        if (cnt == 0) {
            if (!rest && dedicated) { rest = true; continue; }
            break;
        }
        if (a.Pdev && ok) {                              // host -> HBM, block after block (a block's rows are contiguous)
            const int perBlock = 8 * a.ndim, n = cnt * perBlock;
            for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {
                double v[4];
                size_t at[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * kBlock, blk = i / perBlock;
                    at[k] = (size_t)stream_row(x, 8 * (c + blk)) * a.ndim + (size_t)(i - blk * perBlock);
                    v[k] = (i < n && at[k] < (size_t)a.nrows * a.ndim) ? a.P[at[k]] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k * kBlock < n && at[k] < (size_t)a.nrows * a.ndim) a.Pdev[at[k]] = v[k];
            }
            stream_stores_done();
            __syncthreads();
        }
        // the wave's rows of the claim (one per block), one after the other; ONE wait for the L2's acknowledgements, then
        // their stamps (a wait per row put the stores' round trip on the path of every row)
        for (int k = 0; k < cnt; ++k) {
            const int r = stream_row(x, 8 * (c + k) + wave);
            if (r >= a.nrows) continue;
            SampleHdr* hdrp = reinterpret_cast<SampleHdr*>(reinterpret_cast<double*>(a.hdr) + (size_t)r * a.hdr_stride);
            if (ok) {
                setup_sample<kZeroPad>(as, (long)r, lane, a.recs + (size_t)r * a.rec_stride, a.taps + (size_t)r * a.tap_stride, true, hdrp, true);
            } else if (lane == 0) {                      // gave up on the host: a row nobody will mistake for a result
                SampleHdr h;
                h.cont = 0.0; h.bot = 1.0; h.ncl = 0; h.n = 0; h.bad = 1; h.ngeneral = 0;
                *hdrp = h;
            }
        }
        stream_stores_done();                            // the rows' records / taps / headers are in the L2 before their stamps
        if (lane < cnt) {
            const int r = stream_row(x, 8 * (c + lane) + wave);
            if (r < a.nrows) __hip_atomic_store(a.ready + r, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
}

// A workgroup of the streaming launch leaves the kernel: the last one out re-arms the queues for the next launch and
// tells the host (status[1] = gen; every result of every workgroup has been acknowledged by the memory system by
// then, and results and word travel to the host as posted writes in that order -- the host polls this word instead of
// waiting for the stream's signal).  Thread 0 wrote the workgroup's results itself.
__device__ __forceinline__ void stream_exit(const KArgs& a, int tid) {
    if (tid != 0) return;
    stream_stores_done();
    if (atomicAdd(&a.sctl->exited, 1u) == gridDim.x - 1) {
        unsigned lo = ~0u, hi = 0u;                      // (diagnostic: how evenly the dispatcher spread the grid over the XCDs)
        for (int k = 0; k < kXcds; ++k) {
            const unsigned n = __hip_atomic_load(&a.sctl->arrive[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo = min(lo, n); hi = max(hi, n);
        }
        __hip_atomic_store(a.status + 2, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.status + 3, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int k = 0; k < kXcds; ++k) {
            __hip_atomic_store(&a.sctl->arrive[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.sctl->sq_eager[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.sctl->sq_rest[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.sctl->queue[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&a.sctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stream_stores_done();
        __hip_atomic_store(a.status + 1, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

constexpr int kMinWaves = 4;            // waves per SIMD the fused kernel is compiled for (2 workgroups of 8 waves per CU)
// Everything the next work item needs from global memory, requested while the current item is still in its
// convolution / likelihood phase (the loads then have the whole reduction to land in).
struct ItemLoads {
    double treg[(VT_NY * VT_NTOT + kBlock - 1) / kBlock];   // this thread's slice of the universal table T
    double rreg[2];                                         // its slice of the sample's records (covers ncl_cap <= 128)
    double tapreg;                                          // its LSF tap
    double nu[kPpt];                                        // pixel frequencies of the tile
    double nuNode;                                          // this lane's interpolation node
    unsigned long long tileMask;                            // interpolable segments of the tile
    SampleHdr hd;
};

// Issue every global load of work item w (one memory round trip; the record and tap copies run to their
// provisioned sizes, which do not depend on the header: slots beyond the sample's own counts hold stale
// values that are never read).
// kCoh: the streaming launch -- header, records and taps of a live point were written by a wave of the SAME launch on the
// same XCD, into rows of their own 128-byte lines (a.hdr_stride / rec_stride / tap_stride); see stream_setup_phase.
template <bool kZeroPad, bool selfHalo, bool kInline, bool kCoh = false>
__device__ __forceinline__ void request_item(const KArgs& a, int w, int tid, ItemLoads& L) {
    // The thread index is laundered through an empty asm so that the (item-invariant) load addresses are formed
    // here, from one register, instead of being hoisted out of the item loop and kept alive -- ~30 registers --
    // through the component loop, which sits at the kernel's 128-register limit.
    asm volatile("" : "+v"(tid));
    constexpr int kTRegs = (VT_NY * VT_NTOT + kBlock - 1) / kBlock;
    constexpr int kRecRegs = 2;
    const int recTotal = a.ncl_cap * kRecStride, tapTotal = 2 * a.n_cap + 8;
    const int s = w / a.ntiles;
    const int tileIdx = w - s * a.ntiles;
#pragma unroll
    for (int i = 0; i < kTRegs; ++i) {
        const int idx = tid + i * kBlock;
        L.treg[i] = (idx < VT_NY * VT_NTOT) ? a.tabs[idx] : 0.0;
    }
    if (!kInline) {                                      // (one-launch variant: the workgroup sets the live point up itself)
        // (plain, cached loads in both cases.  Streaming launch: the row was set up by a wave of THIS XCD -- its L2 holds
        // what was written -- and owns its 128-byte lines, so this CU's L1 has not seen them before)
        const double* gr = a.recs + (size_t)s * (kCoh ? a.rec_stride : recTotal);
        const double* gt = a.taps + (kCoh ? (size_t)s * a.tap_stride : (a.taps_shared ? 0 : (size_t)s * tapTotal));
        L.hd = kCoh ? *reinterpret_cast<const SampleHdr*>(reinterpret_cast<const double*>(a.hdr) + (size_t)s * a.hdr_stride) : a.hdr[s];
#pragma unroll
        for (int i = 0; i < kRecRegs; ++i) L.rreg[i] = (tid + i * kBlock < recTotal) ? gr[tid + i * kBlock] : 0.0;
        L.tapreg = (tid < tapTotal) ? gt[tid] : 0.0;
    }
    const int t0 = tileIdx * a.tile;
    const int ext0 = selfHalo ? 0 : t0 - a.n_cap;
    // (the pixel count passes through an empty asm: the reciprocal the wrap's `%` needs is then formed here, per
    // item, instead of being hoisted out of the item loop and held -- spilled, in the multi-tile instantiations --
    // through the component loop)
    int npixW = a.npix;
    asm volatile("" : "+s"(npixW));
#pragma unroll
    for (int j = 0; j < kPpt; ++j) {
        int e = ext0 + tid + j * kBlock;
        if (!selfHalo && (e < 0 || e >= a.npix)) {          // (self-halo: nu is padded to the thread count)
            if (kZeroPad) e = 0;                             // jnp.convolve 'same' zero padding (:674)
            else { e %= npixW; if (e < 0) e += npixW; }      // astropy boundary='wrap'
        }
        L.nu[j] = a.nu[e];
    }
    L.nuNode = 0.0;
    L.tileMask = 0;
    if (kFarInterp) {
        const int wv = tid >> 6, ln = tid & 63;
        int e = ext0 + 64 * wv + kBlock * (ln >> 3) + VT_INTERP_NODES[ln & 7];
        if (!selfHalo && (e < 0 || e >= a.npix)) { e %= npixW; if (e < 0) e += npixW; }   // such segments are never interpolated
        L.nuNode = a.nu[e];
        L.tileMask = a.segok[tileIdx];                                         // bit m = segment m = wave + 8 j
    }
}

// PERSISTENT kernel: the grid is the number of workgroup slots of the chip (2 per CU), and every workgroup walks
// over work items w = (live point, pixel tile): its first item is blockIdx.x, the following ones come from an
// atomic queue (a.queue, reset by the set-up kernel of the same launch), so that fast and slow samples balance
// out.  Per item nothing is re-launched: the next item's records / taps / table slices / frequencies are
// requested before the likelihood terms of the current one and written to LDS behind the barrier that ends it.
// Every wave leaves the item loop at the same item count (the queue value is broadcast through LDS), so no wave
// is ever left behind a barrier.
// kInline (small calls -- the one-theta-at-a-time solvers): there is no set-up kernel; wave 0 of the workgroup runs
// setup_sample() for its live point straight into LDS (one launch instead of two on a latency-bound path; every tile of
// a tiled spectrum repeats the set-up, which costs nothing when the chip is empty).
// kStream (the host-pointer entries' large batches): ONE launch for the whole call, no set-up kernel, no copy command.
// The grid sets the live points up itself (stream_setup_phase) while the parameter rows are still arriving in the
// page-locked block the kernel reads them from, and an item goes to the component loop once its row's stamp is there.
template <bool kZeroPad, bool kSelfHalo, int kLinesPerSync, bool kInline, bool kStream>
__device__ __forceinline__ void fused_items(const KArgs& a, double* smem) {
    double* sTab = smem;                                   // 2 x kLinesPerSync folded tables
    double* sRec = sTab + 2 * kLinesPerSync * kTabPad;                     // ncl_cap * 8
    double* sW = sRec + a.ncl_cap * kRecStride;            // taps, zero-padded to a multiple of 8
    double* sRed = sW + (2 * a.n_cap + 8);                 // kRedDoubles: 3 * kWaves partials + the next item index
    double* sWt = sRed + kRedDoubles;                      // [8][64] interpolation weights, node-major
    double* sF = sWt + 64 * VT_INODES;                        // tile_doubles(tile + 2 n_cap)
    int* sNext = reinterpret_cast<int*>(sRed + 3 * kWaves);

    const int tid0 = threadIdx.x;
    // The universal table T lives in the LDS region that later holds the flux tile (T is dead once
    // the component loop ends).  Each thread folds ONE coefficient slot per line.
    double* sT = sF;
    constexpr int kTRegs = (VT_NY * VT_NTOT + kBlock - 1) / kBlock;
    constexpr int kRecRegs = 2;                        // covers ncl_cap <= 128 without a second trip
    constexpr bool selfHalo = kSelfHalo;               // (a.selfhalo chooses the instantiation on the host)
    if (kFarInterp) sWt[(tid0 & 7) * 64 + (tid0 >> 3)] = a.wtab[tid0];      // kBlock == 64 * VT_INODES
    const int recTotal = a.ncl_cap * kRecStride, tapTotal = 2 * a.n_cap + 8;
    // streaming launch: this workgroup's XCD, whose queue hands out LOCAL tickets over the XCD's own live points
    const int xcd = kStream ? xcd_id() : 0;
    const int nItems = kStream ? stream_rows_of(a.nrows, xcd) * a.ntiles : a.nitems;

    // ticket -> work item: with an ordered hand-out, ticket t is tile (t % ntiles) of live point order[t / ntiles]
    constexpr bool kOrdered = kSelfHalo && !kInline && !kStream;   // (the host passes a.order only to these instantiations)
    // ticket -> work item.  Streaming launch: local ticket t of the XCD = tile (t % ntiles) of its local live point t / ntiles
    auto item_of = [&](int t) -> int {
        if (kStream) { const int j = t / a.ntiles; return stream_row(xcd, j) * a.ntiles + (t - j * a.ntiles); }
        return (kOrdered && a.order) ? a.order[t] : t;
    };
    // The next ticket is published to the workgroup behind the component loop -- where the queue's answer (and the
    // order look-up) has long arrived -- rather than before the item's first barrier.  (A streaming launch's queue
    // is shared by all XCDs: published at once, its atomic's round trip to memory sat on every item's path, +7 %.)
    constexpr bool kDeferTicket = kOrdered || kStream;
    int w;
    unsigned int* const queue = kStream ? &a.sctl->queue[xcd] : a.queue;
    // tickets the grid's workgroups start with (the queue continues behind them); the workgroups of a streaming launch
    // -- whose set-up phase is over: see the kernel -- all start from their XCD's queue
    const int firstTickets = kStream ? 0 : (int)gridDim.x;
    if (kStream) {
        if (tid0 == 0) sNext[0] = (int)atomicAdd(queue, 1u);
        __syncthreads();
        const int t = sNext[0];
        __syncthreads();
        if (t >= nItems) {                             // (workgroup-uniform) nothing left for a late-comer
            stream_exit(a, tid0);
            return;
        }
        w = item_of(t);
        stream_wait_ready(a, w / a.ntiles, 0u);
    } else {
        w = item_of(blockIdx.x);                       // grid <= nItems
    }
    ItemLoads L;
    request_item<kZeroPad, kSelfHalo, kInline, kStream>(a, w, tid0, L);

    while (true) {
        // Per item the thread index passes through an empty asm: everything derived from it (LDS offsets, tile
        // positions, global addresses -- dozens of registers) is then formed where it is used instead of being
        // hoisted out of the item loop and kept alive through the component loop, which sits at the kernel's
        // 128-register limit.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const bool hasCoef = tid < VT_NTOT;
        const int coefPos = tid + (tid >= VT_Z0_OFF ? 1 : 0) + (tid >= VT_ZF_OFF ? 1 : 0);
        const bool coreCoef = tid < VT_NCORE;
        const int s = w / a.ntiles;
        const int tileIdx = w - s * a.ntiles;
        // ---- 1. per-sample set-up comes from mcalf_sample_kernel: header, records, taps -----------------
        // The tile always carries the full provisioned halo n_cap (so that its 64-pixel segments are the
        // same for every sample); a sample with a shorter kernel simply starts `shift` entries in.
        //
        // Self-halo mode (a spectrum that fits ONE tile, the usual case): the periodic halo of the convolution
        // consists of copies of the tile's own pixels, so only the npix real pixels are evaluated -- thread index =
        // pixel index, every 64-pixel segment starts at a multiple of 64 and none crosses the seam -- and each
        // flux value is stored at its body position and, near the ends, at its halo position too.  (With the halo
        // evaluated as part of the tile, the segments that contain the seam cannot be interpolated; the three
        // waves that own them then hold every barrier of the component loop back.)
        const int t0 = tileIdx * a.tile;
        const int tlen = min(a.tile, a.npix - t0);
        const int ext0 = selfHalo ? 0 : t0 - a.n_cap;
        const int extCount = tlen + 2 * a.n_cap;
        double nu[kPpt], tau[kPpt];
#pragma unroll
        for (int j = 0; j < kPpt; ++j) {
            const int idx = tid + j * kBlock;
            const int e = ext0 + idx;
            const bool zero = kZeroPad && !selfHalo && (e < 0 || e >= a.npix);
            nu[j] = L.nu[j];
            tau[j] = (zero && idx < extCount) ? INFINITY : 0.0;  // exp(-inf) = 0
        }
        // far-wing interpolation state: this lane's node pixel, the wave's interpolable segments
        const double nuNode = L.nuNode;
        double farNode = 0.0;
        unsigned long long segOk = 0;
        if (kFarInterp) {
            const int wv = tid >> 6;
#pragma unroll
            for (int j = 0; j < kPpt; ++j) segOk |= ((L.tileMask >> (wv + 8 * j)) & 1ULL) << (8 * j);
            segOk = uniform64(segOk);
        }
        SampleHdr hd;
        SampleHdr* sHdr = reinterpret_cast<SampleHdr*>(sRed + 2 * kWaves);   // (one-launch variant; the slot is scratch until the reduction)
        static_assert(sizeof(SampleHdr) <= kWaves * sizeof(double), "header fits the scratch slot");
        if (kInline) {
            if (tid < 64) setup_sample<kZeroPad>(a, s, tid, sRec, sW, true, sHdr, tileIdx == 0);
        } else {
            hd = L.hd;
#pragma unroll
            for (int i = 0; i < kRecRegs; ++i)
                if (tid + i * kBlock < recTotal) sRec[tid + i * kBlock] = L.rreg[i];
            if (recTotal > kRecRegs * kBlock) {
                const double* gr = a.recs + (size_t)s * (kStream ? a.rec_stride : recTotal);
                for (int i = tid + kRecRegs * kBlock; i < recTotal; i += kBlock) sRec[i] = gr[i];
            }
            if (tid < tapTotal) sW[tid] = L.tapreg;
            if (tapTotal > kBlock) {
                const double* gt = a.taps + (kStream ? (size_t)s * a.tap_stride : (a.taps_shared ? 0 : (size_t)s * tapTotal));
                for (int i = tid + kBlock; i < tapTotal; i += kBlock) sW[i] = gt[i];
            }
        }
#pragma unroll
        for (int i = 0; i < kTRegs; ++i) {
            const int idx = tid + i * kBlock;
            if (idx < VT_NY * VT_NTOT) sT[idx] = L.treg[i];
        }
        // the item after this one: the first comes from the grid, the rest from the queue
        // Ticket of the item after this one: the first comes from the grid, the rest from the queue.  With an
        // ordered hand-out (single-tile instantiations) the ticket still has to be looked up in a.order -- a second
        // dependent memory round trip -- so thread 0 keeps both in registers and publishes them behind the
        // component loop, where they have long arrived; otherwise the ticket is published at once.
        int tHeld = nItems, wHeld = 0;
        unsigned stHeld = 0u;
        if (tid == 0) {
            tHeld = a.persist ? firstTickets + (int)atomicAdd(queue, 1u) : nItems;
            if (kOrdered) wHeld = (tHeld < nItems) ? item_of(tHeld) : 0;
            else if (kStream) {
                // (streaming launch: thread 0 also takes a first look at the next row's stamp -- the answer lands while
                // the component loop runs and travels to the workgroup with the ticket)
                wHeld = (tHeld < nItems) ? item_of(tHeld) : 0;
                if (tHeld < nItems) stHeld = __hip_atomic_load(a.ready + wHeld / a.ntiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { sNext[0] = tHeld; sNext[1] = tHeld; }
        }

        // ---- 2. tau for this thread's pixels ----------------------------------------------------
        __syncthreads();                                   // publishes sRec, sW, sT, sNext (and the header of the one-launch variant)
        if (kInline) hd = *sHdr;
        const double cont = hd.cont, bot = hd.bot;
        // (streaming launch: a wait that ran out may leave a header nobody wrote -- keep its counts inside the buffers)
        const int ncl = kStream ? min(max(hd.ncl, 0), a.ncl_cap) : hd.ncl, n = kStream ? min(max(hd.n, 0), a.n_cap) : hd.n;
        const bool bad = hd.bad != 0;
        const int shift = a.n_cap - n;
        int tNext = 0, wNext = 0;
        if (!kDeferTicket) {
            tNext = __builtin_amdgcn_readfirstlane(sNext[0]);            // the next ticket ...
            wNext = __builtin_amdgcn_readfirstlane(sNext[1]);            // ... and the work item it stands for
        }
        int buf = 0;
        const int ncl_run = ncl;
        // kLinesPerSync lines are folded per workgroup barrier (their tables are double-buffered), which
        // halves the barriers and averages the per-wave core/wing imbalance over more work.
        for (int cl0 = 0; cl0 < ncl_run; cl0 += kLinesPerSync) {
            // Wave priority falls as the workgroup progresses, so of the two workgroups sharing a CU the one
            // that is behind gets the issue slots (measured -3.5 % at config B).
            if (4 * cl0 < ncl_run) __builtin_amdgcn_s_setprio(3);
            else if (4 * cl0 < 2 * ncl_run) __builtin_amdgcn_s_setprio(2);
            else if (4 * cl0 < 3 * ncl_run) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            double* tabs = sTab + buf * (kLinesPerSync * kTabPad);
            if (hasCoef) {
                double Tn[VT_NY];
#pragma unroll
                for (int nn = 0; nn < VT_NY; ++nn) Tn[nn] = sT[nn * VT_NTOT + tid];
                // No test per line: past the last record the last one is folded again into a slot nobody reads.
                // The group's Horner chains are written step by step ACROSS the lines, so that they issue
                // interleaved (the fold sits on every wave's path to the barrier; chain after chain it is bound by
                // the latency of 6 dependent FMAs per line).
                double fy[kLinesPerSync], fs[kLinesPerSync], fc[kLinesPerSync];
#pragma unroll
                for (int l = 0; l < kLinesPerSync; ++l) {
                    const double* rec = sRec + min(cl0 + l, ncl_run - 1) * kRecStride;
                    fy[l] = rec[3];
                    fs[l] = coreCoef ? rec[4] : rec[5];
                    fc[l] = Tn[VT_NY - 1];
                }
#pragma unroll
                for (int nn = VT_NY - 2; nn >= 0; --nn) {
#pragma unroll
                    for (int l = 0; l < kLinesPerSync; ++l) fc[l] = fma(fc[l], fy[l], Tn[nn]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int l = 0; l < kLinesPerSync; ++l) tabs[l * kTabPad + coefPos] = fc[l] * fs[l];   // = fold_coef()
            } else if (tid >= kBlock - 64 && tid < kBlock - 64 + 3) {
                // the last wave folds nothing: three of its lanes copy each line's [uthr, A, B] behind its coefficients
                const int which = tid - (kBlock - 64);                                 // 0: uthr, 1: A, 2: B
                const int src = (which == 0) ? 7 : which - 1;
#pragma unroll
                for (int l = 0; l < kLinesPerSync; ++l)
                    tabs[l * kTabPad + kLineLds + which] = sRec[min(cl0 + l, ncl_run - 1) * kRecStride + src];
            }
            __syncthreads();
            buf ^= 1;
            // one copy of the (large) per-line body: keeps the loop inside the instruction cache
            const int lmax = __builtin_amdgcn_readfirstlane(min(kLinesPerSync, ncl_run - cl0));   // (kept scalar)
#pragma unroll 1
            for (int l = 0; l < lmax; ++l) eval_line(tabs + l * kTabPad, nu, tau, nuNode, farNode, segOk);
        }
        if (hd.ngeneral > 0 && ncl_run > 0) eval_general_lines(sRec, ncl, nu, tau);
        __builtin_amdgcn_s_setprio(0);
        // Interpolate the far-wing node sums to the pixels (tau[j] += sum_k W[lane][k] F[segment j][node k]),
        // then flux = exp(-tau) into the LDS tile.  The node sums travel through the (now dead) folded-table
        // region, one 64-entry row per wave; the tile holds only the sample's own halo n (<= n_cap).
        double wrow[VT_INODES];
        double* sFar = sTab + (tid >> 6) * 64;
        if (kFarInterp) {
            __syncthreads();                               // every wave is done reading the folded tables
            sFar[tid & 63] = farNode;
#pragma unroll
            for (int k = 0; k < VT_INODES; ++k) wrow[k] = sWt[k * 64 + (tid & 63)];
        }
        const int extTight = tlen + 2 * n;
        // Tile positions of this thread's pixels: a step of kBlock pixels (a multiple of 8) moves an element by
        // kBlock / 8 slots inside its plane, so ONE position per destination (body, low halo copy, high halo copy)
        // serves all eight pixels as base + 64 j.
        const int posBody = tile_pos(selfHalo ? tid + n : tid - shift);
        const int posLow = tile_pos(tid + n + a.npix), posHigh = tile_pos(tid + n - a.npix);   // (self-halo copies)
        static_assert(kBlock % 8 == 0, "tile_pos(i + kBlock) == tile_pos(i) + kBlock / 8");
        // Two pixels per round: their interpolation sums and exponentials are independent chains the scheduler
        // interleaves (one pixel at a time the phase is bound by the latency of a single ~35-instruction chain).
        // The fences keep it at two: without them the compiler issues the node sums of all eight segments at
        // once and spills them.
        static_assert(kPpt % 2 == 0, "pixels are processed in pairs");
#pragma unroll
        for (int j0 = 0; j0 < kPpt; j0 += 2) {
            double fl[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                double tj = tau[j];
                if (kFarInterp) {
                    double add = 0.0;
#pragma unroll
                    for (int k = 0; k < VT_INODES; ++k) add = fma(wrow[k], sFar[8 * j + k], add);
                    tj += add;
                }
                fl[jj] = exp_neg(tj);                      // :377 (product of exp == exp of sum)
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                if (selfHalo) {
                    const int p = tid + j * kBlock;        // pixel index; tile layout [n halo | npix body | n halo]
                    if (p < a.npix) {
                        sF[posBody + (kBlock / 8) * j] = fl[jj];
                        // periodic copies (astropy boundary='wrap'); the JAX path pads with zeros instead (:674).
                        // Only the first / last pixel groups can hold halo pixels: a scalar test skips the rest.
                        if (j * kBlock < n && p < n) sF[posLow + (kBlock / 8) * j] = kZeroPad ? 0.0 : fl[jj];
                        if ((j + 1) * kBlock > a.npix - n && p >= a.npix - n) sF[posHigh + (kBlock / 8) * j] = kZeroPad ? 0.0 : fl[jj];
                    }
                } else {
                    const int pos = tid + j * kBlock - shift;
                    if (pos >= 0 && pos < extTight) sF[posBody + (kBlock / 8) * j] = fl[jj];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
        }
        if (tid < kTileSlack) {
            double zero = 0.0;
            asm volatile("" : "+v"(zero));                 // formed here: hoisted out of the item loop it was spilled
            sF[tile_pos(extTight + tid)] = zero;
        }
        if (kDeferTicket && tid == 0) { sNext[0] = tHeld; sNext[1] = wHeld; if (kStream) sNext[2] = (int)stHeld; }
        __syncthreads();
        if (kDeferTicket) {
            tNext = __builtin_amdgcn_readfirstlane(sNext[0]);
            wNext = __builtin_amdgcn_readfirstlane(sNext[1]);
        }
        // Streaming launch: thread 0's look at the next row's stamp (taken while the component loop ran; the set-up runs
        // far ahead of the queue, so it normally says "set up" and nobody has to ask memory again)
        unsigned stampNext = 0u;
        if (kStream) stampNext = (unsigned)__builtin_amdgcn_readfirstlane(sNext[2]);

        // ---- 3+4. convolution, continuum, likelihood terms -------------------------------------
        // Register sliding window: this thread owns outputs base..base+7; per tap one new flux value
        // and one (broadcast) weight are read from LDS for eight FMAs.
        double acc = 0.0, nnz = 0.0, c4 = 0.0, c5 = 0.0;
        const int base = 8 * tid;
        const bool reduces = (a.mode == kModeLogL || a.mode == kModeChi2);
        if (base < tlen) {
            // The data of this thread's 8 pixels are requested now and consumed after the convolution (the
            // device arrays carry 8 doubles of padding, so the 64-byte reads never need a bounds test).
            // The data of this thread's 8 pixels are requested now and consumed after the convolution (the
            // device arrays carry 8 doubles of padding, so the 64-byte reads never need a bounds test).  The loads
            // are unconditional on purpose -- model-only calls simply ignore them: defined under `if (reduces)`
            // the 24 values become phi(undef, load) ranges that the register allocator of the persistent loop
            // spills one load at a time.
            const size_t o0 = (size_t)(t0 + base);
            double ob[8], is2[8], lg[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) { ob[m] = a.obj[o0 + m]; is2[m] = a.ispec2[o0 + m]; lg[m] = a.lgis[o0 + m]; }
            double win[8], top[8];
            const double* fp = sF + tid;                   // element 8 tid + 8 c + r  ->  fp[r * kPlaneStride + c]
#pragma unroll
            for (int m = 0; m < 8; ++m) { win[m] = fp[m * kPlaneStride]; top[m] = 0.0; }
            const double* wp = sW;
            const int ntaps = 2 * n + 1;
            for (int q0 = 0; q0 + 8 <= ntaps; q0 += 8) {   // whole groups of eight taps
                ++fp;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const double wgt = wp[r];
#pragma unroll
                    for (int m = 0; m < 8; ++m) top[m] = fma(win[(m + r) & 7], wgt, top[m]);
                    win[r] = fp[r * kPlaneStride];         // element base + q0 + r + 8
                }
                wp += 8;
            }
            {                                              // the last 1..7 taps (2n+1 is odd): no zero-weight padding taps
                const int rem = ntaps & 7;                 // wave-uniform
                ++fp;
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    if (r >= rem) break;
                    const double wgt = wp[r];
#pragma unroll
                    for (int m = 0; m < 8; ++m) top[m] = fma(win[(m + r) & 7], wgt, top[m]);
                    win[r] = fp[r * kPlaneStride];
                }
            }
            const double ibot = 1.0 / bot;
            // The plain log-likelihood (no model output, no asymmetric veto, numpy boundary) gets its own loop: in
            // the general one below every pixel drags the mode / veto / output tests along as selects and reloads
            // of spilled scalars (~30 vector instructions per pixel against ~12 here).  Same arithmetic, same order.
            const bool plainLogL = !kZeroPad && a.mode == kModeLogL && !a.asymm && a.model == nullptr;
            if (plainLogL) {
                if (!bad) {                                      // (bad: every term is NaN and is dropped)
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        double mval = top[m] * ibot;
                        mval *= cont;                                                          // :447
                        const double d = ob[m] - mval;
                        double term = is2[m] * (d * d);
                        term = (term - lg[m]) + a.log2pi;                                      // :294
                        acc += (base + m < tlen && !isnan(term)) ? term : 0.0;                 // np.nansum
                    }
                }
            } else if (!kZeroPad && a.mode != kModeLogL && a.mode != kModeChi2 && a.model != nullptr) {
                // Model output alone (reconstruct_spec / reconstruct_onecomp for a batch, numpy boundary): eight
                // consecutive pixels per thread, 64 contiguous bytes, nothing else -- same arithmetic as above.
                double* mrow = a.model + (size_t)s * a.npix + t0 + base;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    double mval = top[m] * ibot;
                    mval *= cont;                                                              // :447
                    if (bad) mval = NAN;
                    if (base + m < tlen) mrow[m] = mval;
                }
            } else
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = base + m;
                const bool live = i < tlen;
                const int pix = t0 + i;
                double mval = kZeroPad ? top[m] : top[m] * ibot;
                if (kZeroPad && (pix < n || pix >= a.npix - n)) mval = sF[tile_pos(min(i, tlen - 1) + n)];   // :677-681 edge reset
                mval *= cont;                                                                  // :447 / :683
                if (bad) mval = NAN;
                if (a.model && live) a.model[(size_t)s * a.npix + pix] = mval;
                if (reduces) {
                    const double d = ob[m] - mval;
                    double term = is2[m] * (d * d);
                    if (a.mode == kModeLogL) term = (term - lg[m]) + a.log2pi;                 // :294
                    if (live && !isnan(term)) acc += term;                                     // np.nansum
                    if (a.mode == kModeChi2 && live && mval != 0.0) nnz += 1.0;      // only chi2 asks whether the model is all zero (:241)
                    if (a.asymm) {                                                             // :298-302 (rare: loaded here)
                        const double resid = d / a.err[o0 + m];
                        if (live && resid > 4.0) c4 += 1.0;
                        if (live && resid > 5.0) c5 += 1.0;
                    }
                }
            }
        }
        const bool more = tNext < nItems;
        // The next item's global loads go out here (the pixel data of this item are consumed, so the registers
        // are free): they land while the reduction and the barrier that ends the item run.
        __builtin_amdgcn_sched_barrier(0);
        // (unconditional -- the last item of a workgroup re-requests a valid item it never uses -- so that the
        // loads REDEFINE every register of L: behind a condition the old values would have to stay alive through
        // the whole item for the merge)
        if (kStream && more) stream_wait_ready(a, wNext / a.ntiles, stampNext);
        if (!kInline) request_item<kZeroPad, kSelfHalo, kInline, kStream>(a, more ? wNext : w, tid, L);   // (one-launch variant: one item per workgroup)
        __builtin_amdgcn_sched_barrier(0);
        if (reduces) {
            acc = wave_sum_to_last(acc);
            if (a.mode == kModeChi2) nnz = wave_sum_to_last(nnz);
            const int wave = tid >> 6;
            if ((tid & 63) == 63) { sRed[wave] = acc; sRed[kWaves + wave] = nnz; }
        }
        double t4 = 0.0, t5 = 0.0;
        if (reduces && a.asymm) {                        // rare path: two more workgroup sums
            __syncthreads();
            t4 = block_sum(c4, sRed + 2 * kWaves, tid);
            t5 = block_sum(c5, sRed + 2 * kWaves, tid);
        }
        __syncthreads();                                 // every wave is past its reads of the flux tile and the taps
        if (reduces && tid == 0) {
            double ssum = 0.0, scnt = 0.0;
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) { ssum += sRed[wv]; scnt += sRed[kWaves + wv]; }
            // LSF wider than the provisioned halo: the model was not computed (the reference would build a longer
            // kernel); the row must not look like a valid likelihood -> logL = -inf, chi2 = +inf
            if (bad) { ssum = INFINITY; scnt = 1.0; }
            if (a.ntiles == 1) {
                const double val = finalize_value(a.mode, ssum, scnt, a.asymm != 0, t4, t5, a.veto4, a.veto5);
                // (streaming launch: the result goes to page-locked host memory and the host reads it as soon as the
                // launch's completion word says so, ahead of the end-of-kernel cache write-back: a system-scope store,
                // written through -- plain stores were seen to linger in one XCD's L2 past the completion word)
                // (one-launch variant: its results go to page-locked memory too, and the host -- or, for the resident
                // kernel below, the next request -- reads them while the kernel is still there)
                if (kStream || kInline) __hip_atomic_store(a.out + s, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else a.out[s] = val;
            } else {
                double* pr = a.partial + ((size_t)s * a.ntiles + tileIdx) * 4;
                pr[0] = ssum; pr[1] = scnt; pr[2] = t4; pr[3] = t5;
            }
        }
        if (kInline || !more) break;                     // wave-uniform: every wave of the workgroup leaves here
        w = wNext;
    }
    if (kStream) stream_exit(a, tid0);
}

template <bool kZeroPad, bool kSelfHalo, int kLinesPerSync, bool kInline, bool kStream>
__global__ __launch_bounds__(kBlock, kMinWaves) void mcalf_fused_kernel(const KArgs a) {
    static_assert(!(kInline && kStream), "the one-launch variant of small calls has no queue to stream through");
    extern __shared__ __align__(16) double smem[];
    if constexpr (kStream) {
        stream_setup_phase<kZeroPad>(a, threadIdx.x, xcd_id(), reinterpret_cast<int*>(smem));

        // The item loop reads its arguments afresh from the kernel-argument segment (through a pointer the compiler
        // cannot see through): their live ranges then start HERE, as in the two-kernel variant.  With one set of
        // values alive across both phases the set-up's scalar-register pressure spilled the loop's arguments for
        // their whole life (183 scalar spills, ~500 more v_readlane reloads on every item's path).
        // (a typed copy out of the constant address space: pointers loaded from there are known to be global, so the loop
        // keeps its global_load / global_atomic instructions -- copied word by word they became generic pointers, and a
        // FLAT load also counts as an LDS operation: every LDS wait of the loop then waited for HBM)
        typedef __attribute__((address_space(4))) const KArgs ArgSeg;
        ArgSeg* kp = (ArgSeg*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        KArgs fresh = *(const KArgs*)kp;                  // (an aggregate copy: the compiler splits it into scalar loads of typed fields)
        fused_items<kZeroPad, kSelfHalo, kLinesPerSync, kInline, kStream>(fresh, smem);
    } else {
        fused_items<kZeroPad, kSelfHalo, kLinesPerSync, kInline, kStream>(a, smem);
    }
}

// RESIDENT one-theta evaluator (opt-in: mcalf_set_resident).  The solvers call the likelihood one theta at a time
// (lnlhood_pc / _dy / _mn, hires_fitter.py:250-285), and of such a call's 18 us only 11 are the kernel: the rest is the
// launch -- and launches of different processes serialise at 8.5 us each.  So ONE workgroup stays on the chip between calls
// and takes its requests from a page-locked mailbox: the host writes the row and bumps `req`; thread 0 polls `req` (a PCIe
// read per look), the workgroup copies the row into LDS with system-scope loads (nothing of a request is ever read
// through a cache), runs the one-launch variant's item on it -- same code, same bits -- and the result goes out as a
// system-scope store, followed by `ack`.  The kernel LEAVES after `idle_ticks` without a request (or when told to):
// state = leaving, one more look at `req` (a request that slipped in is served, state = running again), state = gone.
// The host treats "gone" as "launch another one"; a request posted behind the last look is therefore never lost, and no
// wave can stay behind: every wait is bounded by the idle limit.
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
template <bool kZeroPad, bool kSelfHalo>
__global__ __launch_bounds__(kBlock, 2) void mcalf_resident_kernel(const KArgs a, ResidentBox* boxes, ResidentShared* shared,
                                                                           long long idle_ticks, int row_offset_doubles) {
    extern __shared__ __align__(16) double smem[];
    double* sRow = smem + row_offset_doubles;            // behind everything the item uses
    unsigned* sCtl = reinterpret_cast<unsigned*>(sRow + kResRowMax);
    const int tid = threadIdx.x;
    ResidentBox* box = boxes + blockIdx.x;
    const unsigned long long t_launch = __builtin_amdgcn_s_memrealtime();
    // the number of the last request this mailbox has had answered: what comes next is new
    unsigned seen = __hip_atomic_load(&box->ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    while (true) {
        if (tid == 0) {
            unsigned r;
            while (true) {
                // (`req` and `quit` share eight bytes: ONE PCIe read per look)
                const unsigned long long both = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&box->req), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_SYSTEM);
                r = (unsigned)both;
                if (r != seen) break;
                bool leave = (unsigned)(both >> 32) != 0u || __hip_atomic_load(&shared->leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (!leave && blockIdx.x == 0) {         // workgroup 0 keeps the launch's clock
                    const unsigned long long last = __hip_atomic_load(&shared->last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long since = last > t_launch ? last : t_launch;
                    if ((long long)(__builtin_amdgcn_s_memrealtime() - since) > idle_ticks) leave = true;
                }
                if (leave) {
                    __hip_atomic_store(&shared->leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&box->state, kResLeaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    stream_stores_done();                // (a read does not pass the posted write: the host has "leaving" before this look)
                    r = __hip_atomic_load(&box->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (r != seen) { __hip_atomic_store(&box->state, kResRunning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __hip_atomic_store(&box->state, kResGone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            sCtl[0] = r;
        }
        __syncthreads();
        const unsigned r = sCtl[0];
        __syncthreads();
        if (r == seen) return;                           // (workgroup-uniform) gone
        // the row, past every cache
        if (tid < kResRowMax) {
            const unsigned long long bits = __hip_atomic_load(reinterpret_cast<unsigned long long*>(box->row) + tid, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_SYSTEM);
            sRow[tid] = __builtin_bit_cast(double, bits);
        }
        __syncthreads();
        // (the arguments are read afresh from the kernel-argument segment for every request, as in the streaming launch:
        // kept alive across the waiting loop they cost the item 150 scalar spills)
        typedef __attribute__((address_space(4))) const KArgs ArgSeg;
        ArgSeg* kp = (ArgSeg*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        KArgs b = *(const KArgs*)kp;
        // (the item of workgroup k is "live point k": its row is the one in LDS, its result slot the mailbox's)
        const int rowlen = b.ndim;
        b.P = sRow - (size_t)blockIdx.x * rowlen;
        b.out = &box->result - blockIdx.x;
        fused_items<kZeroPad, kSelfHalo, 4, true, false>(b, smem);
        if (tid == 0) {                                  // (thread 0 stored the result itself)
            stream_stores_done();
            __hip_atomic_store(&box->ack, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_max(&shared->last, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        seen = r;
        __syncthreads();
    }
}

__global__ void mcalf_finalize_kernel(const double* partial, double* out, long batch, int ntiles, int mode,
                                      int asymm, double veto4, double veto5) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    double sum = 0.0, cnt = 0.0, c4 = 0.0, c5 = 0.0;
    for (int t = 0; t < ntiles; ++t) {
        const double* pr = partial + (s * ntiles + t) * 4;
        sum += pr[0]; cnt += pr[1]; c4 += pr[2]; c5 += pr[3];
    }
    out[s] = finalize_value(mode, sum, cnt, asymm != 0, c4, c5, veto4, veto5);
}

__global__ void mcalf_hjert_kernel(const double* x, const double* y, long n, double* out, const double* tabs,
                                   int node_form) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hjert_folded(x[i], y[i], tabs, node_form != 0);
}

__global__ void mcalf_scale_cube_kernel(const double* lo, const double* hi, const double* cube, long total,
                                        int ndim, int slot, int int_ncomp, double* theta) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % ndim);
    // separately rounded multiply and add (no FMA contraction), as numpy evaluates
    // cube*ptp + min (hires_fitter.py:206 / :214)
    double v;
    {
#pragma clang fp contract(off)
        const double scaled = cube[i] * (hi[d] - lo[d]);
        v = scaled + lo[d];
    }
    if (int_ncomp && d == slot) v = trunc(v);        // :207-208
    theta[i] = v;
}

}  // namespace mcalf

// ==========================================================================================
// Host side: context + C ABI
// ==========================================================================================
using namespace mcalf;

#define MCALF_STR_(x) #x
#define MCALF_STR(x) MCALF_STR_(x)
static thread_local std::string g_last_error;
constexpr int kMaxChunks = 8;

struct mcalf_ctx {
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
    int host_plan_n = 0;                       // host-pointer entry (0 = the built-in plans)
    int num_cu = 256;
    int persist = 1;                    // fused kernel as a persistent grid (MCALF_PERSIST=0: one workgroup per item)
    int setup_block = 512;              // threads per workgroup of the set-up kernel (MCALF_SETUP_BLOCK: 64 .. 512)
    unsigned int* d_queue = nullptr;    // [kMaxChunks] work-item queues of the persistent kernel
    int* d_order = nullptr;             // [batch] hand-out order of the persistent kernel (per row block)
    size_t cap_order = 0;
    int ordered = 1;                    // MCALF_ORDER=0: hand the live points out in row order
    hipStream_t aux[kMaxChunks - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxChunks - 1] = {};
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
    bool fail_preflight = false;            // MCALF_TEST_FAIL_PREFLIGHT=1: tests inject a workspace-growth failure
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
    unsigned int stream_gen = 0;
    int stream_on = 1;                      // MCALF_STREAM: 0 = the row-block pipeline of round 2 instead; 1 = automatic (spectra that
                                            // fit one tile: measured, config E's five tiles per live point run 1.3 % faster through
                                            // the pipeline); 2 = always
    int stream_wgs = 16;                    // MCALF_STREAM_WGS: workgroups dedicated to the set-up while rows are outstanding
    int stream_eager = 0;                   // MCALF_STREAM_EAGER: blocks of 8 rows per XCD any workgroup may set up (0: what the first items need)
    int stream_chunk = 32;                  // MCALF_STREAM_CHUNK: rows such a workgroup claims (and copies to HBM) at a time
    int stream_trace = 0;                   // MCALF_STREAM_TRACE=1 (diagnostic): host-side time per phase of the streaming entry
    int stream_device = 0;                  // MCALF_STREAM_DEVICE=1 (diagnostic): the *_device scalar entries take the streaming launch too
    int stream_poll = 1;                    // MCALF_STREAM_POLL=0: wait for the stream's signal instead of polling h_ctl[1]
    double stream_timeout_s = 0.5;          // MCALF_STREAM_TIMEOUT: longest wait of a wave inside the kernel
    mcalf_launch_info_t last = {};      // what the last call did (mcalf_last_launch)
};
constexpr int kCtlWords = 64, kCtlArrived = 16;

static int set_err(mcalf_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return set_err(ctx, MCALF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

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

template <typename T>
static int grow(mcalf_ctx* ctx, T** ptr, size_t* cap, size_t need_elems) {
    if (need_elems <= *cap) return MCALF_OK;
    if (*ptr) HIP_TRY(ctx, hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)ptr, need_elems * sizeof(T)));
    *cap = need_elems;
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
extern "C" const char* mcalf_version(void) { return "mcalf_hip 0.3 (gfx950, abi " MCALF_STR(MCALF_ABI_VERSION) ") src " MCALF_SRC_HASH; }

extern "C" const char* mcalf_last_error(const mcalf_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

static void comm_release(mcalf_ctx* ctx);

// MCALF_STREAM_TRACE=1 (diagnostic): mean host-side microseconds per phase of the streaming entry, printed when the
// context is destroyed
struct StreamTrace { double t[6] = {}; long n = 0; };
static StreamTrace g_stream_trace;
static inline double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static void stream_trace_report(const mcalf_ctx* ctx);
static void resident_stop(mcalf_ctx* ctx);

extern "C" void mcalf_destroy(mcalf_ctx* ctx) {
    if (!ctx) return;
    stream_trace_report(ctx);
    (void)hipSetDevice(ctx->device);
    comm_release(ctx);
    resident_stop(ctx);
    if (ctx->res_stream) (void)hipStreamDestroy(ctx->res_stream);
    if (ctx->h_box) (void)hipHostFree((void*)ctx->h_box);
    if (ctx->d_res_shared) (void)hipFree(ctx->d_res_shared);
    void* bufs[] = {ctx->d_nu, ctx->d_obj, ctx->d_ispec2, ctx->d_lgis, ctx->d_err, ctx->d_tabs, ctx->d_lines, ctx->d_wtab, ctx->d_segok,
                    ctx->d_P,  ctx->d_out, ctx->d_partial, ctx->d_model, ctx->d_bounds, ctx->d_prior, ctx->d_recs, ctx->d_taps, ctx->d_hdr,
                    ctx->d_queue, ctx->d_order, ctx->d_sws};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->h_ctl) (void)hipHostFree((void*)ctx->h_ctl);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
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

// The instantiations of the fused kernel: (JAX semantics, self-halo tile, lines per barrier) for batches, and the
// one-launch variant (set-up inside the kernel, always 4 lines per barrier) for small calls.
static const void* fused_kernel_ptr(bool jax, bool selfhalo, int lps, bool inl = false, bool stream = false) {
#define MCALF_K(J, S, L, I) reinterpret_cast<const void*>(&mcalf_fused_kernel<J, S, L, I, false>)
#define MCALF_KS(J, S, L) reinterpret_cast<const void*>(&mcalf_fused_kernel<J, S, L, false, true>)
    if (stream) {
        if (lps == 5) {
            if (jax) return selfhalo ? MCALF_KS(true, true, 5) : MCALF_KS(true, false, 5);
            return selfhalo ? MCALF_KS(false, true, 5) : MCALF_KS(false, false, 5);
        }
        if (jax) return selfhalo ? MCALF_KS(true, true, 4) : MCALF_KS(true, false, 4);
        return selfhalo ? MCALF_KS(false, true, 4) : MCALF_KS(false, false, 4);
    }
#undef MCALF_KS
    if (inl) {
        if (jax) return selfhalo ? MCALF_K(true, true, 4, true) : MCALF_K(true, false, 4, true);
        return selfhalo ? MCALF_K(false, true, 4, true) : MCALF_K(false, false, 4, true);
    }
    if (lps == 5) {
        if (jax) return selfhalo ? MCALF_K(true, true, 5, false) : MCALF_K(true, false, 5, false);
        return selfhalo ? MCALF_K(false, true, 5, false) : MCALF_K(false, false, 5, false);
    }
    if (jax) return selfhalo ? MCALF_K(true, true, 4, false) : MCALF_K(true, false, 4, false);
    return selfhalo ? MCALF_K(false, true, 4, false) : MCALF_K(false, false, 4, false);
#undef MCALF_K
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
    if (const char* e = std::getenv("MCALF_LINES_PER_SYNC")) {
        const int v = std::atoi(e);
        if (v == 4 || (v == 5 && ext_for(5) == ext_for(4))) ctx->lps = v;
    }
    const size_t fixed_doubles = fixed_for(ctx->lps);
    size_t ext = ext_for(ctx->lps);
    if (ext < 2 * (size_t)ctx->n_cap + 64)
        return set_err(ctx, MCALF_ERR_RANGE,
                       "LSF half-width %d px (specres_max %.3g km/s at %.3g km/s/px) with %d component-lines does not "
                       "fit a %d-pixel workgroup tile / the %zu-byte LDS budget", ctx->n_cap, rmax, sp->velstep,
                       ctx->ncl_cap, kExtMax, kLdsBudget);
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
    const void* kernels[] = {fused_kernel_ptr(false, false, 4), fused_kernel_ptr(false, true, 4),
                             fused_kernel_ptr(true, false, 4),  fused_kernel_ptr(true, true, 4),
                             fused_kernel_ptr(false, false, 5), fused_kernel_ptr(false, true, 5),
                             fused_kernel_ptr(true, false, 5),  fused_kernel_ptr(true, true, 5),
                             fused_kernel_ptr(false, false, 4, true), fused_kernel_ptr(false, true, 4, true),
                             fused_kernel_ptr(true, false, 4, true),  fused_kernel_ptr(true, true, 4, true),
                             fused_kernel_ptr(false, false, 4, false, true), fused_kernel_ptr(false, true, 4, false, true),
                             fused_kernel_ptr(true, false, 4, false, true),  fused_kernel_ptr(true, true, 4, false, true),
                             fused_kernel_ptr(false, false, 5, false, true), fused_kernel_ptr(false, true, 5, false, true),
                             fused_kernel_ptr(true, false, 5, false, true),  fused_kernel_ptr(true, true, 5, false, true)};
    for (const void* k : kernels)
        HIP_TRY(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_queue, kMaxChunks * sizeof(unsigned int)));
        HIP_TRY(ctx, hipMemset(ctx->d_queue, 0, kMaxChunks * sizeof(unsigned int)));
        const char* pe = std::getenv("MCALF_PERSIST");
        if (pe && *pe) ctx->persist = std::atoi(pe) != 0;
        if (const char* oe = std::getenv("MCALF_ORDER")) ctx->ordered = std::atoi(oe) != 0;
        ctx->inline_max_items = 2 * ctx->num_cu;                    // launches that fit the chip in one round of workgroups
        if (const char* ie = std::getenv("MCALF_INLINE_MAX")) ctx->inline_max_items = std::max(0, std::atoi(ie));
        if (const char* re = std::getenv("MCALF_RESIDENT_US")) ctx->resident_us = std::min(1000000, std::max(0, std::atoi(re)));
        if (const char* sb = std::getenv("MCALF_SETUP_BLOCK")) {      // diagnostic: geometry of the set-up kernel
            const int v = std::atoi(sb);
            if (v >= 64 && v <= kSetupBlockMax && v % 64 == 0) ctx->setup_block = v;
        }
        if (const char* hp = std::getenv("MCALF_HOST_PLAN")) {      // e.g. "1,3,4": relative block sizes (diagnostic)
            int n = 0;
            for (const char* q = hp; *q && n < kMaxChunks;) {
                const int v = std::atoi(q);
                if (v > 0) ctx->host_plan[n++] = v;
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
            if (n > 0) ctx->host_plan_n = n;
        }
        if (const char* fp = std::getenv("MCALF_TEST_FAIL_PREFLIGHT")) ctx->fail_preflight = std::atoi(fp) != 0;
        if (const char* e = std::getenv("MCALF_STREAM")) ctx->stream_on = std::min(std::max(std::atoi(e), 0), 2);
        if (const char* e = std::getenv("MCALF_STREAM_WGS")) ctx->stream_wgs = std::min(std::max(std::atoi(e), 1), ctx->num_cu);
        if (const char* e = std::getenv("MCALF_STREAM_POLL")) ctx->stream_poll = std::atoi(e) != 0;
        if (const char* e = std::getenv("MCALF_STREAM_EAGER")) ctx->stream_eager = std::max(std::atoi(e), 0);
        if (const char* e = std::getenv("MCALF_STREAM_CHUNK")) ctx->stream_chunk = std::min(std::max(std::atoi(e) & ~7, 8), 512);
        if (const char* e = std::getenv("MCALF_STREAM_DEVICE")) ctx->stream_device = std::atoi(e);
        if (const char* e = std::getenv("MCALF_STREAM_TRACE")) ctx->stream_trace = std::atoi(e) != 0;
        if (const char* e = std::getenv("MCALF_STREAM_TIMEOUT")) { const double v = std::atof(e); if (v > 0.0 && v <= 60.0) ctx->stream_timeout_s = v; }

        const char* env = std::getenv("MCALF_CHUNKS");            // 0 / unset: automatic; n: exactly n row blocks
        if (env && *env) {
            const int v = std::atoi(env);
            if (v >= 0 && v <= kMaxChunks) ctx->chunks_req = v;
        }
    }
    return MCALF_OK;
}

extern "C" int mcalf_create(const mcalf_spec* spec, mcalf_ctx** out) {
    if (!out) return set_err(nullptr, MCALF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    mcalf_ctx* ctx = new (std::nothrow) mcalf_ctx();
    if (!ctx) return set_err(nullptr, MCALF_ERR_NOMEM, "out of host memory");
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
    info->n_cap = ctx->n_cap;
    info->tile = ctx->tile;
    info->ntiles = ctx->ntiles;
    info->device = ctx->device;
    info->npix = ctx->npix;
    snprintf(info->arch, sizeof info->arch, "%s", ctx->arch.c_str());
    return MCALF_OK;
}

extern "C" int mcalf_last_launch(const mcalf_ctx* ctx, mcalf_launch_info_t* info) {
    if (!ctx || !info) return set_err(nullptr, MCALF_ERR_INVALID, "NULL argument");
    *info = ctx->last;
    return MCALF_OK;
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = grow(ctx, &ctx->d_P, &ctx->cap_P, (size_t)batch * std::max(ctx->ndim, 5)))) return rc;
    if ((rc = grow(ctx, &ctx->d_out, &ctx->cap_out, (size_t)batch))) return rc;
    if ((rc = grow(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)batch * ctx->ntiles * 4))) return rc;
    if ((rc = grow_sample_ws(ctx, batch))) return rc;
    return MCALF_OK;
}

// Enqueue rows [row0, row0 + nrows) of a batch on `stream`: set-up kernel, fused kernel, finalize when tiled.
// `chunk` selects the row of the shared tap table this block writes and reads (fixed-resolution contexts).
// `from_cube`: dP holds unit-cube rows, mapped through the prior box while decoding; d_theta (optional)
// receives the transformed rows.
// The kernel arguments of rows [row0, row0 + nrows) of a batch (everything but the launch geometry).
static KArgs make_kargs(const mcalf_ctx* ctx, int mode, const double* dP, int64_t row0, int64_t nrows, int chunk,
                        int targonly, int onecomp_fill, double* d_out, double* d_model, bool from_cube, double* d_theta) {
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
    return a;
}

static int launch_range(mcalf_ctx* ctx, int mode, const double* dP, int64_t row0, int64_t nrows, int chunk,
                        int targonly, int onecomp_fill, double* d_out, double* d_model, hipStream_t stream,
                        bool from_cube, double* d_theta, bool timed_ok) {
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    KArgs a = make_kargs(ctx, mode, dP, row0, nrows, chunk, targonly, onecomp_fill, d_out, d_model, from_cube, d_theta);
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
        if (ctx->conv_mode == MCALF_CONV_SAME_EDGE_JAX)
            hipLaunchKernelGGL(mcalf_sample_kernel<true>, sgrid, sblock, 0, stream, a, (long)nrows);
        else
            hipLaunchKernelGGL(mcalf_sample_kernel<false>, sgrid, sblock, 0, stream, a, (long)nrows);
        HIP_TRY(ctx, hipGetLastError());
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
    if (reduces && ctx->ntiles > 1) {
        const int fb = 256;
        hipLaunchKernelGGL(mcalf_finalize_kernel, dim3((unsigned)((nrows + fb - 1) / fb)), dim3(fb), 0, stream,
                           a.partial, a.out, (long)nrows, ctx->ntiles, mode, a.asymm, a.veto4, a.veto5);
        HIP_TRY(ctx, hipGetLastError());
    }
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

static int ensure_aux(mcalf_ctx* ctx, int naux) {
    if (!ctx->ev_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < naux; ++i) {
        if (!ctx->aux[i]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
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
static int launch_preflight(mcalf_ctx* ctx, int mode, int64_t batch) {
    if (batch == 0) return MCALF_OK;
    if (batch < 0 || batch * (int64_t)ctx->ntiles > 0x7fff0000LL)
        return set_err(ctx, MCALF_ERR_RANGE, "batch %lld too large", (long long)batch);
    const bool reduces = (mode == kModeLogL || mode == kModeChi2);
    int rc;
    if (ctx->fail_preflight) return set_err(ctx, MCALF_ERR_NOMEM, "workspace growth failed (injected by MCALF_TEST_FAIL_PREFLIGHT)");
    if (reduces && ctx->ntiles > 1 && (rc = grow(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)batch * ctx->ntiles * 4)))
        return rc;
    return grow_sample_ws(ctx, batch);
}

static void stream_trace_report(const mcalf_ctx* ctx) {
    if (!ctx->stream_trace || g_stream_trace.n == 0) return;
    const double n = (double)g_stream_trace.n;
    std::fprintf(stderr, "mcalf stream trace (%ld calls, us per call; workgroups per XCD in the last launch: %u .. %u): pointer checks %.2f, "
                 "prepare %.2f, launch %.2f, stage rows %.2f, wait %.2f, copy out %.2f\n", g_stream_trace.n,
                 ctx->h_ctl ? ctx->h_ctl[2] : 0u, ctx->h_ctl ? ctx->h_ctl[3] : 0u, g_stream_trace.t[0] / n, g_stream_trace.t[1] / n, g_stream_trace.t[2] / n,
                 g_stream_trace.t[3] / n, g_stream_trace.t[4] / n, g_stream_trace.t[5] / n);
    g_stream_trace = StreamTrace();
}

static int stream_prepare(mcalf_ctx* ctx, int mode, int64_t batch);
static int stream_launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, double* d_out, hipStream_t stream, int wgs,
                         int64_t eager_rows, bool staged, bool host_rows, bool from_cube = false);
static bool stream_qualifies(const mcalf_ctx* ctx, int64_t batch);

static int launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, int targonly, int onecomp_fill,
                  double* d_out, double* d_model, hipStream_t stream, bool from_cube = false,
                  double* d_theta = nullptr) {
    if (batch == 0) return MCALF_OK;
    int rc;
    // MCALF_STREAM_DEVICE=1 (diagnostic): device-pointer batches through the streaming single launch as well -- every
    // row is there from the start, so the whole grid sets up eight live points per workgroup and goes on to the items
    if (ctx->stream_device && (mode == kModeLogL || mode == kModeChi2) && !from_cube && !d_model && ctx->chunks_req <= 1 &&
        stream_qualifies(ctx, batch)) {
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
    return (ctx && batch > 0) ? pick_chunks(ctx, batch) : 0;
}

extern "C" int mcalf_set_chunks(mcalf_ctx* ctx, int32_t nchunks) {
    if (!ctx || nchunks < 0 || nchunks > kMaxChunks)
        return set_err(ctx, MCALF_ERR_INVALID, "nchunks must be 0 (automatic) .. %d", kMaxChunks);
    ctx->chunks_req = nchunks;
    return MCALF_OK;
}

extern "C" int mcalf_profile_begin(mcalf_ctx* ctx, int32_t max_launches) {
    if (!ctx || max_launches <= 0) return set_err(ctx, MCALF_ERR_INVALID, "bad arguments");
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->last.path = MCALF_PATH_DEVICE; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    return launch(ctx, kModeLogL, dP, batch, 0, 0, dlogL, nullptr, (hipStream_t)stream);
}

extern "C" int mcalf_model_batch_device(mcalf_ctx* ctx, const double* dP, int64_t batch, int32_t targonly,
                                        double* dflux, void* stream) {
    if (!ctx || (batch > 0 && (!dP || !dflux))) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->last.path = MCALF_PATH_DEVICE; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    return launch(ctx, kModeModel, dP, batch, targonly ? 1 : 0, 0, nullptr, dflux, (hipStream_t)stream);
}

constexpr size_t kSmallDoubles = 65536;     // up to 512 KB of parameters (and as many results) go the zero-copy way

// True when `p` is page-locked host memory the copy engines can read / write directly.
static bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();                          // ordinary pageable memory: not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// Large scalar-output batches through host pointers: the rows are cut into blocks that alternate between two
// streams, each block being  H2D of its parameter rows -> set-up + fused kernels -> D2H of its results,  so the
// PCIe traffic and the per-block set-up of block k+1 run under the kernels of block k.  Pageable caller memory
// is staged through a page-locked block of the context (the host copies block k+1 in while the GPU works on
// block k); page-locked caller memory is used by the copy engines directly.
static int run_host_pipelined(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly,
                              int fill, double* out_scalar) {
    int rc;
    // Row blocks: an explicit request gives equal blocks; the automatic plan is a SMALL first block (the GPU starts
    // after one eighth of the input has arrived) followed by larger ones (large launches run the persistent grid
    // and leave fewer tails).  Measured on MI355X, config C (device-resident 0.251 ms per batch): pageable input
    // 1:1:2:4 0.311 ms, 1:3:4 0.319, 2:6 0.320, four equal blocks 0.333, one block 0.351; page-locked input (no
    // staging copy on the host thread) 1:7 0.297, 2:6 0.301, 1:1:2:4 0.307, four equal blocks 0.321.
    const bool pin_in = is_pinned_host(P), pin_out = is_pinned_host(out_scalar);
    int weights[kMaxChunks];
    int nchunks = 0;
    if (ctx->chunks_req > 0) {
        for (nchunks = 0; nchunks < ctx->chunks_req && nchunks < kMaxChunks; ++nchunks) weights[nchunks] = 1;
    } else if (ctx->host_plan_n > 0) {
        for (nchunks = 0; nchunks < ctx->host_plan_n; ++nchunks) weights[nchunks] = ctx->host_plan[nchunks];
    } else if (pin_in) {
        weights[0] = 1; weights[1] = 7; nchunks = 2;
    } else {
        weights[0] = 1; weights[1] = 1; weights[2] = 2; weights[3] = 4; nchunks = 4;
    }
    if ((int64_t)nchunks > batch) nchunks = (int)batch;
    if (ctx->profiling) nchunks = 1;
    int64_t bounds[kMaxChunks + 1];
    {
        int total = 0, run = 0;
        for (int c = 0; c < nchunks; ++c) total += weights[c];
        bounds[0] = 0;
        for (int c = 0; c < nchunks; ++c) {
            run += weights[c];
            bounds[c + 1] = batch * run / total;
        }
    }
    if ((rc = ensure_aux(ctx, 1))) return rc;
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
    // A failure in block k leaves blocks < k in flight on both streams, reading the staging block / the caller's
    // page-locked rows and writing the caller's results: never return under them (the next call may free the
    // staging block, the caller its arrays).  Every error below therefore leaves through `fail`.
    hipError_t he = hipSuccess;
    const char* what = "";
    rc = MCALF_OK;
    for (int c = 0; c < nchunks && rc == MCALF_OK && he == hipSuccess; ++c) {
        const int64_t r0 = bounds[c], n = bounds[c + 1] - r0;
        if (n == 0) continue;
        hipStream_t st = streams[c & 1];
        const double* src = P + (size_t)r0 * rowlen;
        if (!pin_in) {
            std::memcpy(stage_in + (size_t)r0 * rowlen, src, (size_t)n * rowlen * sizeof(double));
            src = stage_in + (size_t)r0 * rowlen;
        }
        he = hipMemcpyAsync(ctx->d_P + (size_t)r0 * rowlen, src, (size_t)n * rowlen * sizeof(double), hipMemcpyHostToDevice, st);
        if (he != hipSuccess) { what = "H2D copy of a row block"; break; }
        rc = launch_range(ctx, mode, ctx->d_P, r0, n, c, targonly, fill, d_stage_out ? d_stage_out : ctx->d_out, nullptr, st,
                          false, nullptr, nchunks == 1);
        if (rc != MCALF_OK) break;
        if (!d_stage_out) {
            he = hipMemcpyAsync(stage_out + r0, ctx->d_out + r0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
            if (he != hipSuccess) { what = "D2H copy of a result block"; break; }
        }
    }
    const hipError_t s0 = hipStreamSynchronize(ctx->stream);
    const hipError_t s1 = (nchunks > 1) ? hipStreamSynchronize(ctx->aux[0]) : hipSuccess;
    if (rc != MCALF_OK) return rc;                                   // (message set by launch_range)
    if (he != hipSuccess) return set_err(ctx, MCALF_ERR_HIP, "%s failed: %s", what, hipGetErrorString(he));
    if (s0 != hipSuccess || s1 != hipSuccess)
        return set_err(ctx, MCALF_ERR_HIP, "stream synchronisation failed: %s", hipGetErrorString(s0 != hipSuccess ? s0 : s1));
    if (!pin_out) std::memcpy(out_scalar, stage_out, (size_t)batch * sizeof(double));
    return MCALF_OK;
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
// Workspaces of the streaming launch for `batch` live points, and the words it shares with the host.
static int stream_prepare(mcalf_ctx* ctx, int mode, int64_t batch) {
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
        ctx->stream_gen = 0;
    }
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
static int stream_launch(mcalf_ctx* ctx, int mode, const double* dP, int64_t batch, double* d_out, hipStream_t stream, int wgs,
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
    if ((mode == kModeLogL || mode == kModeChi2) && ctx->ntiles > 1) {
        const int fb = 256;
        hipLaunchKernelGGL(mcalf_finalize_kernel, dim3((unsigned)((batch + fb - 1) / fb)), dim3(fb), 0, stream,
                           a.partial, a.out, (long)batch, ctx->ntiles, mode, a.asymm, a.veto4, a.veto5);
        HIP_TRY(ctx, hipGetLastError());
    }
    return MCALF_OK;
}

static bool stream_qualifies(const mcalf_ctx* ctx, int64_t batch) {
    const int64_t slots = 2LL * ctx->num_cu, nitems = batch * ctx->ntiles;
    return ctx->persist && nitems >= 4 * slots && nitems <= 0x7fff0000LL && batch <= 0x7fff0000LL;
}

// theta = cube * (hi - lo) + lo with the separately rounded multiply and add numpy performs (hires_fitter.py:206 / :214)
// and int() on the ncomp slot (:207-208): the arithmetic of sample_param() / mcalf_scale_cube_kernel, on the host -- the
// host-pointer cube entries form the rows they hand back while their launch runs.
static void host_scale_cube(const mcalf_ctx* ctx, const double* cube, int64_t batch, double* theta) {
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

static int run_host_stream(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, double* out_scalar, bool* taken,
                           bool from_cube = false, double* theta_out = nullptr) {
    *taken = false;
    const bool trace = ctx->stream_trace;
    double tm[7] = {};
    if (trace) tm[0] = now_us();
    if (!ctx->stream_on || ctx->chunks_req > 0 || ctx->host_plan_n > 0 || ctx->profiling || !stream_qualifies(ctx, batch)) return MCALF_OK;
    if (ctx->stream_on == 1 && ctx->ntiles > 1) return MCALF_OK;
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
    if (he == hipSuccess && ctx->stream_poll && !tiled) {
        const unsigned int want = ctx->stream_gen;
        for (unsigned long spins = 0;; ++spins) {
            if (__atomic_load_n(const_cast<unsigned int*>(ctx->h_ctl + 1), __ATOMIC_ACQUIRE) == want) { polled = true; break; }
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
    if (ctx->h_ctl[0] != 0u) {
        // a wave ran out of patience (the grid has drained by now): not an answer -- the caller goes the pipelined way
        (void)hipStreamSynchronize(ctx->stream);
        HIP_TRY(ctx, hipMemset(ctx->d_sctl, 0, sizeof(StreamCtl)));
        HIP_TRY(ctx, hipDeviceSynchronize());
        return MCALF_OK;
    }
    if (trace) tm[5] = now_us();
    if (!pin_out) std::memcpy(out_scalar, stage_out, (size_t)batch * sizeof(double));
    if (trace) {
        tm[6] = now_us();
        for (int k = 0; k < 6; ++k) g_stream_trace.t[k] += tm[k + 1] - tm[k];
        g_stream_trace.n++;
    }
    *taken = true;
    return MCALF_OK;
}

// What a result slot of the small-call block holds until its kernel has written it: a quiet NaN with a payload no
// arithmetic produces (the kernels' NaNs are the canonical one or carry an operand's payload).
constexpr uint64_t kResultPending = 0x7FF8C0DEC0DE0001ull;

// The page-locked, device-mapped block small calls go through: parameters in its first half, results in its second.
static int ensure_small(mcalf_ctx* ctx) {
    if (!ctx->h_small) {
        // (coherent: the host fills the result slots before a launch and reads them while it runs)
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_small, 2 * kSmallDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->d_small, ctx->h_small, 0));
    }
    return MCALF_OK;
}

// ---- resident one-theta evaluator, host side (device side: mcalf_resident_kernel) -------------------------------------
static const void* resident_kernel_ptr(bool jax, bool selfhalo) {
    if (jax) return selfhalo ? reinterpret_cast<const void*>(&mcalf_resident_kernel<true, true>)
                             : reinterpret_cast<const void*>(&mcalf_resident_kernel<true, false>);
    return selfhalo ? reinterpret_cast<const void*>(&mcalf_resident_kernel<false, true>)
                    : reinterpret_cast<const void*>(&mcalf_resident_kernel<false, false>);
}

// Tell the resident kernel to leave and wait until it has (bounded by its own idle limit).
static void resident_stop(mcalf_ctx* ctx) {
    if (!ctx->h_box || !ctx->res_stream) return;
    if (ctx->res_alive) {
        __atomic_store_n(&ctx->h_box->quit, 1u, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(ctx->res_stream);
        ctx->res_alive = false;
    }
}

static bool resident_serves(const mcalf_ctx* ctx, int mode, int64_t batch, int rowlen, bool from_cube) {
    return ctx->resident_us > 0 && mode == kModeLogL && batch == 1 && ctx->ntiles == 1 && rowlen <= kResRowMax && !from_cube &&
           !ctx->profiling;
}

// One theta through the resident kernel: post the request (launching the kernel when there is none), spin on the result.
static int resident_call(mcalf_ctx* ctx, const double* row, int rowlen, double* out) {
    if (!ctx->h_box) {
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_box, sizeof(ResidentBox), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset((void*)ctx->h_box, 0, sizeof(ResidentBox));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->d_box, (void*)ctx->h_box, 0));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->res_stream, hipStreamNonBlocking));
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
        if (__atomic_load_n(res, __ATOMIC_ACQUIRE) != kResultPending) break;
        if (__atomic_load_n(&box->state, __ATOMIC_ACQUIRE) == kResGone) {
            // the kernel left without having seen this request (it looks once more after saying "leaving", so a request it
            // has seen is answered): a new one takes it
            if (__atomic_load_n(res, __ATOMIC_ACQUIRE) != kResultPending) break;
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (idle_us == 0) resident_stop(ctx);
    ctx->resident_us = idle_us;
    return MCALF_OK;
}

// Small scalar-output calls (up to kSmallDoubles parameters: single-theta calls, config B's batch), zero-copy: a
// single-theta call is dominated by the latency of its two copy commands.  from_cube / theta_out: as in run_host_stream.
static int run_host_small(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly, int fill,
                          double* out_scalar, bool from_cube, double* theta_out) {
    int rc;
    if (resident_serves(ctx, mode, batch, rowlen, from_cube)) return resident_call(ctx, P, rowlen, out_scalar);
    if ((rc = ensure_small(ctx))) return rc;
    std::memcpy(ctx->h_small, P, (size_t)batch * rowlen * sizeof(double));
    ctx->last.path = MCALF_PATH_HOST_ZEROCOPY; ctx->last.pinned_in = ctx->last.pinned_out = 0;
    // Completion is read off the results: their slots are filled with a NaN no kernel produces, and the call is over
    // when none is left -- a stream wait costs an interrupt and a thread wake-up on top of the kernel, a fifth of a
    // one-theta call.  (The stream is asked now and then, so that a failed launch cannot keep the call here.)
    uint64_t* res = reinterpret_cast<uint64_t*>(ctx->h_small + kSmallDoubles);
    const bool poll = ctx->stream_poll != 0;
    if (poll)
        for (int64_t i = 0; i < batch; ++i) __atomic_store_n(res + i, kResultPending, __ATOMIC_RELEASE);
    rc = launch(ctx, mode, ctx->d_small, batch, targonly, fill, ctx->d_small + kSmallDoubles, nullptr, ctx->stream, from_cube);
    if (rc) return rc;
    if (theta_out) host_scale_cube(ctx, P, batch, theta_out);       // (under the launch)
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
    std::memcpy(out_scalar, ctx->h_small + kSmallDoubles, (size_t)batch * sizeof(double));
    return MCALF_OK;
}

// Host-pointer entries: stage through the context's workspaces on its private stream.
static int run_host(mcalf_ctx* ctx, int mode, const double* P, int64_t batch, int rowlen, int targonly, int fill,
                    double* out_scalar, double* out_model) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (batch < 0) return set_err(ctx, MCALF_ERR_INVALID, "negative batch");
    if (batch == 0) return MCALF_OK;
    if (!P || (!out_scalar && !out_model)) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if (out_scalar && !out_model && (size_t)batch * rowlen <= kSmallDoubles && (size_t)batch <= kSmallDoubles)
        return run_host_small(ctx, mode, P, batch, rowlen, targonly, fill, out_scalar, false, nullptr);
    if ((rc = grow(ctx, &ctx->d_P, &ctx->cap_P, (size_t)batch * rowlen))) return rc;
    if (out_scalar && (rc = grow(ctx, &ctx->d_out, &ctx->cap_out, (size_t)batch))) return rc;
    if (out_model && (rc = grow(ctx, &ctx->d_model, &ctx->cap_model, (size_t)batch * ctx->npix))) return rc;
    if (out_scalar && !out_model) {
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
extern "C" int mcalf_broker_serve_resident(mcalf_ctx* ctx, void* boxes, int32_t slots, volatile uint64_t* stop, int32_t idle_us,
                                           uint64_t* stats, double max_seconds) {
    static_assert(sizeof(ResidentBox) == MCALF_MAILBOX_BYTES, "mailbox layout of include/mcalf_hip.h");
    static_assert(kResultPending == MCALF_RESULT_PENDING, "pending pattern of include/mcalf_hip.h");
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (!boxes || !stop || slots < 1 || slots > 256 || idle_us < 1 || idle_us > 1000000 || (reinterpret_cast<uintptr_t>(boxes) & 63))
        return set_err(ctx, MCALF_ERR_INVALID, "resident broker: 1 .. 256 mailboxes at a 64-byte aligned address, idle limit 1 .. 1000000 us");
    if (ctx->ntiles != 1 || ctx->ndim > kResRowMax)
        return set_err(ctx, MCALF_ERR_RANGE, "resident broker: the spectrum must fit one pixel tile and a row 64 parameters "
                       "(%d tiles, %d parameters): use mcalf_broker_serve", ctx->ntiles, ctx->ndim);
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
    if (const char* over = std::getenv("MCALF_RCCL_LIB")) {
        if (*over && !(h = dlopen(over, RTLD_NOW | RTLD_LOCAL)))
            return set_err(ctx, MCALF_ERR_COMM, "MCALF_RCCL_LIB=%s: %s", over, dlerror());
    }
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

static void comm_release(mcalf_ctx* ctx) {
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < 2; ++k)
        if (ctx->ev_comm_used[k]) HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_comm[k], 0));
    return MCALF_OK;
}

static int comm_ensure_streams(mcalf_ctx* ctx) {
    if (!ctx->comm_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
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
        g_last_error = ctx->err;
    }
    return rc_local;
}

extern "C" int mcalf_loglike_gather_device(mcalf_ctx* ctx, const double* dP, int64_t batch_local, double* dlogL_local,
                                           double* dlogL_all, int32_t root, void* stream) {
    return mcalf_loglike_gatherv_device(ctx, dP, batch_local, dlogL_local, dlogL_all, nullptr, root, stream);
}

extern "C" int mcalf_set_prior(mcalf_ctx* ctx, const double* lo, const double* hi, int32_t int_ncomp) {
    if (!ctx) return set_err(nullptr, MCALF_ERR_INVALID, "ctx is NULL");
    if (!lo || !hi) return set_err(ctx, MCALF_ERR_INVALID, "NULL argument");
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t total = (size_t)batch * ctx->ndim;
    int rc;
    // the paths of mcalf_loglike_batch, with the prior transform applied while the rows are decoded and the transformed
    // rows formed on the host under the launch: the zero-copy small call, ONE streaming launch for large batches
    if (total <= kSmallDoubles && (size_t)batch <= kSmallDoubles)
        return run_host_small(ctx, kModeLogL, cube, batch, ctx->ndim, 0, 0, logL, true, theta);
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
    hipLaunchKernelGGL(mcalf_scale_cube_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->d_bounds, ctx->d_bounds + ctx->ndim, ctx->d_P, (long)total, ctx->ndim, ctx->startind,
                       int_ncomp, ctx->d_model);
    HIP_TRY(ctx, hipGetLastError());
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
            hipLaunchKernelGGL(mcalf_hjert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, (long)n,
                               dout, dtabs, node_form);
            e = hipGetLastError();
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
