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
// This file is the DEVICE side: every kernel, and at its end the table of kernel entry points the host files launch
// through (kernel_args.h declares it together with the argument block).  The C ABI (include/mcalf_hip.h) lives in
// host_abi.cpp (contexts, launches, the entries), host_stream.cpp (the streaming launch of the host-pointer entries),
// broker.cpp (resident evaluator, likelihood broker) and comm.cpp (RCCL gather).
#include <hip/hip_runtime.h>

#include <cmath>

#include "kernel_args.h"
#include "voigt_device.h"

namespace mcalf {

// The XCD a wave runs on (XCC_ID, bits 3:0 of hardware register 20 on gfx942 / gfx950): all four bits for the context's
// probe (mcalf_xcd_probe_kernel), three for the streaming launch -- the host only takes that launch on a stream whose
// workgroups the probe saw on exactly the XCDs 0 .. 7 of an unpartitioned MI355X (host_stream.cpp), and checks after every
// launch that each of them did receive workgroups (status[2]).
__device__ __forceinline__ int xcd_raw() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15); }
__device__ __forceinline__ int xcd_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (kXcds - 1)); }

__device__ __forceinline__ int tile_pos(int i) { return (i & 7) * kPlaneStride + (i >> 3); }

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
            __hip_atomic_store(a.status, kStreamHostLate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
                __hip_atomic_store(a.status, kStreamStampLate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    constexpr int nx = kXcds;
    const int nloc = stream_rows_of(a.nrows, x, nx), nblk = (nloc + 7) >> 3;      // this XCD's live points / local blocks
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
                const int last = min(stream_row(x, 8 * (c + cnt) - 1, nx), a.nrows - 1);
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
                    at[k] = (size_t)stream_row(x, 8 * (c + blk), nx) * a.ndim + (size_t)(i - blk * perBlock);
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
            const int r = stream_row(x, 8 * (c + k) + wave, nx);
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
            const int r = stream_row(x, 8 * (c + lane) + wave, nx);
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
        // status[2] / [3]: the fewest / most workgroups an XCD received.  An XCD's rows are set up and evaluated by ITS
        // workgroups only: with status[2] == 0 one of them got none (a partition mode or CU mask the context's probe did
        // not reflect) and its result slots are untouched -- the host then does not take the launch for an answer
        // (run_host_stream fails over to the row-block pipeline).
        unsigned lo = ~0u, hi = 0u;
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
    const int nItems = kStream ? stream_rows_of(a.nrows, xcd, kXcds) * a.ntiles : a.nitems;

    // ticket -> work item: with an ordered hand-out, ticket t is tile (t % ntiles) of live point order[t / ntiles]
    constexpr bool kOrdered = kSelfHalo && !kInline && !kStream;   // (the host passes a.order only to these instantiations)
    // ticket -> work item.  Streaming launch: local ticket t of the XCD = tile (t % ntiles) of its local live point t / ntiles
    auto item_of = [&](int t) -> int {
        if (kStream) { const int j = t / a.ntiles; return stream_row(xcd, j, kXcds) * a.ntiles + (t - j * a.ntiles); }
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
        // (streaming instantiations: continuum and tap sum are the same in every lane -- say so, and they ride through the
        // component loop in scalar registers; as vector values the tiled instantiation spilled them to scratch.  The batch
        // instantiations are left as they were compiled in round 4: their register allocation is what was measured.)
        const double cont = kStream ? __builtin_bit_cast(double, uniform64(__builtin_bit_cast(unsigned long long, hd.cont))) : hd.cont;
        const double bot = kStream ? __builtin_bit_cast(double, uniform64(__builtin_bit_cast(unsigned long long, hd.bot))) : hd.bot;
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
// (mailbox and shared words: ResidentBox / ResidentShared in kernel_args.h)
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

// done_word != nullptr (the streaming launch of a TILED spectrum): `out` is page-locked host memory the host reads as soon as
// *done_word == gen -- results go out as system-scope stores, and the last workgroup to finish (fin_count, re-armed by
// it) writes the word behind them, as stream_exit() does for single-tile spectra.
__global__ void mcalf_finalize_kernel(const double* partial, double* out, long batch, int ntiles, int mode,
                                      int asymm, double veto4, double veto5, unsigned int* fin_count, unsigned int* done_word,
                                      unsigned int gen) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < batch) {
        double sum = 0.0, cnt = 0.0, c4 = 0.0, c5 = 0.0;
        for (int t = 0; t < ntiles; ++t) {
            const double* pr = partial + (s * ntiles + t) * 4;
            sum += pr[0]; cnt += pr[1]; c4 += pr[2]; c5 += pr[3];
        }
        const double val = finalize_value(mode, sum, cnt, asymm != 0, c4, c5, veto4, veto5);
        if (done_word) __hip_atomic_store(out + s, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else out[s] = val;
    }
    if (done_word) {                                         // (grid-uniform)
        stream_stores_done();
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(fin_count, 1u) == gridDim.x - 1) {
            __hip_atomic_store(fin_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stream_stores_done();
            __hip_atomic_store(done_word, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
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

// ---- LSF wider than a workgroup tile: two more kernels behind the fused one ---------------------------------------------
// The fused kernel convolves inside a 4096-pixel LDS tile, halo included.  A context whose LSF half-width does not fit
// one (a very finely sampled spectrum, a very coarse resolution: the reference simply builds a longer kernel,
// hires_fitter.py:458-464) runs the SAME fused kernel without its convolution and continuum -- the host passes it
// velstep = 1e300 (hires_fitter.py:445: no convolution while R <= velstep; JAX semantics: a kernel grid of half-width 0,
// i.e. the single tap 1), a fixed continuum of 1 -- into a buffer of
// unconvolved spectra [rows][npix], and these two kernels do the rest: the taps of every live point (any half-width up
// to the one provisioned from specres_max), then the periodic convolution, the continuum and the likelihood terms.
// The convolution is the fused kernel's scheme at a larger scale: a workgroup owns kWideBlockPix consecutive outputs,
// eight per thread, and walks over the taps in passes of kWideTapChunk -- per pass the flux window (outputs + taps of the
// pass, wrapped round the spectrum as often as it takes) and the taps are staged in LDS, the window in the mod-8 planar
// layout, and every thread slides an 8-register window over it: one LDS flux read and one broadcast weight per 8 FMAs.
// Every output accumulates its taps in tap order (tap 0 first), which is astropy's loop order and the order `bot` is
// formed in -- an absorber-free model therefore comes out as exactly 1.0, as it does from the fused kernel.
constexpr int kWideBlock = kWideBlockThreads;
constexpr int kWidePlane = (kWideBlockPix + kWideTapChunk + 16) / 8 + 2;      // 324: plane stride of the staged window
static_assert(kWideBlockPix == 8 * kWideBlock && kWideTapChunk % 8 == 0 && kWidePlane % 32 == 4, "wide convolution geometry");
__device__ __forceinline__ int wide_pos(int i) { return (i & 7) * kWidePlane + (i >> 3); }
__device__ __forceinline__ double wide_block_sum(double v, double* red, int tid) {      // fixed order (deterministic)
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kWideBlock / 64; ++w) s += red[w];
    __syncthreads();
    return s;
}

// Decode (R, continuum) of live point s as setup_sample() does (hires_fitter.py:412-425; mode OneComp: the row is (R, cont, N, z, b)).
__device__ __forceinline__ void wide_decode(const KArgs& a, long s, double& R, double& cont) {
    const int rowlen = (a.mode == kModeOneComp) ? 5 : a.ndim;
    const double* p = a.P + (size_t)s * rowlen;
    if (a.mode == kModeOneComp) { R = p[0]; cont = p[1]; return; }
    R = a.freespecres ? sample_param(a, p, 0) : a.specres_fixed;
    cont = a.freecont ? sample_param(a, p, a.freespecres ? 1 : 0) : a.contval_fixed;
}

// One workgroup per live point: its normalised Gaussian taps w[0 .. 2n] (astropy's Gaussian1DKernel divided by its sum),
// the sum `bot` astropy's loop divides by -- added up tap after tap, tap 0 first, the order the convolution's numerator
// chain runs in (as setup_sample() does for the fused kernel) -- and the header (continuum, bot, n, bad).
// kZeroPad (JAX semantics, hires_fitter.py:549-560,667-670): the kernel grid is FIXED at the context's half-width (from
// the largest resolution), the taps are exp(-k^2 / 2 sigma^2) divided by their sum and nothing else divides (bot = 1).
template <bool kZeroPad>
__global__ __launch_bounds__(kWideBlock) void mcalf_wide_taps_kernel(const KArgs a, double* taps, long tap_stride, SampleHdr* hdr) {
    __shared__ double red[kWideBlock / 64];
    const long s = blockIdx.x;
    const int tid = threadIdx.x;
    double R, cont;
    wide_decode(a, s, R, cont);
    const double sigma = (R / kFwhmToSigma) / a.velstep;         // :454
    long n = 0;
    bool bad = false;
    if (kZeroPad) {
        n = a.jax_half;                                          // :549-560 fixed grid
    } else if (R > a.velstep) {                                  // :445
        const double nd = ceil(kKernelReach * sigma);            // :458
        if (!(nd <= (double)a.n_cap)) bad = true;                // (beyond the half-width provisioned from specres_max; NaN too)
        else n = (long)nd;
    }
    double* w = taps + (size_t)s * tap_stride;
    // (the expressions of setup_sample(), so that a kernel that fits a tile and one that does not form their taps alike)
    const double inv2s2 = kZeroPad ? 1.0 / (2.0 * sigma * sigma) : 0.5 / (sigma * sigma);
    const double amp = kZeroPad ? 1.0 : 1.0 / (sqrt(2.0 * M_PI) * sigma);
    double part = 0.0;
    for (long k = tid; k <= 2 * n; k += kWideBlock) {
        const double dk = (double)(k - n);
        const double g = (n == 0 && !kZeroPad) ? 1.0 : exp_neg((dk * dk) * inv2s2) * amp;
        w[k] = g;
        part += g;
    }
    const double gsum = wide_block_sum(part, red, tid);
    for (long k = tid; k <= 2 * n; k += kWideBlock) w[k] = w[k] / gsum;     // normalize_kernel=True
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        double bot = 0.0;                                        // tap order: a dependent chain of 2n + 1 additions, once per live point
        if (kZeroPad) bot = 1.0;                                 // (:670 normalises the kernel; jnp.convolve divides by nothing)
        else for (long k = 0; k <= 2 * n; ++k) bot += w[k];
        SampleHdr h;
        h.cont = cont; h.bot = bot; h.ncl = 0; h.n = (int)n; h.bad = bad ? 1 : 0; h.ngeneral = 0;
        hdr[s] = h;
    }
}

// Workgroup (blockIdx.x, blockIdx.y) = pixels [2048 x, 2048 x + 2048) of live point y:  model = (sum_k w_k flux[(i + k - n) mod
// npix]) / bot x continuum (astropy boundary='wrap', taps in window order; hires_fitter.py:463-464, :447), then the model
// row and / or the likelihood terms (:292-303) summed over the block into partial[y][x][4], which mcalf_finalize_kernel adds up.
// kZeroPad (JAX semantics): jnp.convolve(model, kernel, 'same') -- zeros outside the spectrum instead of the periodic
// window (:674) -- and the first / last n pixels reset to the unconvolved model (:677-681); nothing divides the sum.
template <bool kZeroPad>
__global__ __launch_bounds__(kWideBlock) void mcalf_wide_conv_kernel(const KArgs a, const double* flux, const double* taps, long tap_stride,
                                                                         const SampleHdr* hdr, int nblocks) {
    __shared__ double red[kWideBlock / 64];
    __shared__ double sF[8 * kWidePlane];
    __shared__ double sW[kWideTapChunk];
    const long s = blockIdx.y;
    const int tid = threadIdx.x;
    const long i0 = (long)blockIdx.x * kWideBlockPix;            // first output pixel of the workgroup
    const SampleHdr h = hdr[s];
    const long n = h.n, npix = a.npix, ntaps = 2 * n + 1;
    const double* f = flux + (size_t)s * npix;
    const double* w = taps + (size_t)s * tap_stride;
    double top[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) top[m] = 0.0;
    if (!h.bad) {                                                // (workgroup-uniform)
        for (long k0 = 0; k0 < ntaps; k0 += kWideTapChunk) {
            const int kt = (int)min((long)kWideTapChunk, ntaps - k0);       // taps of this pass
            // window element e of the pass = flux[(i0 - n + k0 + e) mod npix]: output 8 t + m reads element 8 t + m + k at tap k0 + k
            if (kZeroPad) {
                const long first = i0 - n + k0;
                for (int e = tid; e < kWideBlockPix + kWideTapChunk + 16; e += kWideBlock) {
                    const long j = first + e;
                    sF[wide_pos(e)] = (e < kWideBlockPix + kt && j >= 0 && j < npix) ? f[j] : 0.0;      // :674 zero padding
                }
            } else {
                long src = (i0 - n + k0 + tid) % npix;
                if (src < 0) src += npix;
                const long step = kWideBlock % npix;
                for (int e = tid; e < kWideBlockPix + kWideTapChunk + 16; e += kWideBlock) {
                    sF[wide_pos(e)] = (e < kWideBlockPix + kt) ? f[src] : 0.0;   // (past the pass's window: never multiplied by a tap)
                    src += step;
                    if (src >= npix) src -= npix;
                }
            }
            for (int k = tid; k < kWideTapChunk; k += kWideBlock) sW[k] = (k < kt) ? w[k0 + k] : 0.0;
            __syncthreads();
            double win[8];
            const double* fp = sF + tid;                         // element 8 tid + 8 c + r  ->  fp[r * kWidePlane + c]
#pragma unroll
            for (int m = 0; m < 8; ++m) win[m] = fp[m * kWidePlane];
            const double* wp = sW;
            for (int q0 = 0; q0 + 8 <= kt; q0 += 8) {            // whole groups of eight taps
                ++fp;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const double wgt = wp[r];
#pragma unroll
                    for (int m = 0; m < 8; ++m) top[m] = fma(win[(m + r) & 7], wgt, top[m]);
                    win[r] = fp[r * kWidePlane];
                }
                wp += 8;
            }
            {                                                    // the pass's last 0..7 taps: no zero-weight padding taps
                const int rem = kt & 7;                          // (workgroup-uniform)
                ++fp;
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    if (r >= rem) break;
                    const double wgt = wp[r];
#pragma unroll
                    for (int m = 0; m < 8; ++m) top[m] = fma(win[(m + r) & 7], wgt, top[m]);
                    win[r] = fp[r * kWidePlane];
                }
            }
            __syncthreads();                                     // every wave is past its reads before the next pass overwrites
        }
    }
    const bool reduces = a.mode == kModeLogL || a.mode == kModeChi2;
    double acc = 0.0, nnz = 0.0, c4 = 0.0, c5 = 0.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const long i = i0 + 8 * tid + m;
        if (i >= npix) break;
        double mval = NAN;
        if (!h.bad) {
            mval = kZeroPad ? top[m] : top[m] / h.bot;
            if (kZeroPad && (i < n || i >= npix - n)) mval = f[i];       // :677-681 edge reset to the unconvolved model
            mval *= h.cont;                                      // :447 / :683
        }
        if (a.model) a.model[(size_t)s * npix + i] = mval;
        if (reduces) {
            const double d = a.obj[i] - mval;
            double term = a.ispec2[i] * (d * d);
            if (a.mode == kModeLogL) term = (term - a.lgis[i]) + a.log2pi;      // :294
            if (!isnan(term)) acc += term;                                    // np.nansum
            if (a.mode == kModeChi2 && mval != 0.0) nnz += 1.0;               // :241
            if (a.asymm) {                                                     // :298-302
                const double resid = d / a.err[i];
                if (resid > 4.0) c4 += 1.0;
                if (resid > 5.0) c5 += 1.0;
            }
        }
    }
    if (!reduces) return;                                        // (workgroup-uniform)
    acc = wide_block_sum(acc, red, tid);
    nnz = wide_block_sum(nnz, red, tid);
    if (a.asymm) { c4 = wide_block_sum(c4, red, tid); c5 = wide_block_sum(c5, red, tid); }
    if (tid == 0) {
        if (h.bad) { acc = INFINITY; nnz = 1.0; }               // not computed: logL = -inf, chi2 = +inf (as in the fused kernel)
        double* pr = a.partial + ((size_t)s * nblocks + blockIdx.x) * 4;
        pr[0] = acc; pr[1] = nnz; pr[2] = c4; pr[3] = c5;
    }
}

// Single-component rows (R, cont, N, z, b) with the continuum set to 1: what the fused kernel of a wide-LSF context gets
// (it must deliver the UNCONVOLVED, continuum-free spectrum; mode OneComp reads the continuum from the row itself).
__global__ void mcalf_wide_rows_kernel(const double* rows, double* out, long batch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * 5) return;
    out[i] = (i % 5 == 1) ? 1.0 : rows[i];
}

// One bit per XCD that runs a workgroup of the launch (the host launches a few workgroups per CU on the context's
// stream: mcalf_create, mcalf_set_cu_mask): the XCDs a streaming launch on that stream can count on.
__global__ void mcalf_xcd_probe_kernel(unsigned int* mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << xcd_raw());
}

// ---- kernel entry points for the host files (kernel_args.h) -------------------------------------------------------------
#define MCALF_K(J, S, L, I, T) reinterpret_cast<const void*>(&mcalf_fused_kernel<J, S, L, I, T>)
static const void* const kFusedKernels[] = {
    // [group: batch 4 / batch 5 / one-launch / streaming 4 / streaming 5 lines per barrier][jax][selfhalo]
    MCALF_K(false, false, 4, false, false), MCALF_K(false, true, 4, false, false), MCALF_K(true, false, 4, false, false), MCALF_K(true, true, 4, false, false),
    MCALF_K(false, false, 5, false, false), MCALF_K(false, true, 5, false, false), MCALF_K(true, false, 5, false, false), MCALF_K(true, true, 5, false, false),
    MCALF_K(false, false, 4, true, false),  MCALF_K(false, true, 4, true, false),  MCALF_K(true, false, 4, true, false),  MCALF_K(true, true, 4, true, false),
    MCALF_K(false, false, 4, false, true),  MCALF_K(false, true, 4, false, true),  MCALF_K(true, false, 4, false, true),  MCALF_K(true, true, 4, false, true),
    MCALF_K(false, false, 5, false, true),  MCALF_K(false, true, 5, false, true),  MCALF_K(true, false, 5, false, true),  MCALF_K(true, true, 5, false, true),
};
#undef MCALF_K
const void* fused_kernel_ptr(bool jax, bool selfhalo, int lps, bool inl, bool stream) {
    const int group = stream ? (lps == 5 ? 4 : 3) : inl ? 2 : (lps == 5 ? 1 : 0);
    return kFusedKernels[4 * group + (jax ? 2 : 0) + (selfhalo ? 1 : 0)];
}
int fused_kernel_count() { return (int)(sizeof(kFusedKernels) / sizeof(kFusedKernels[0])); }
const void* fused_kernel_at(int i) { return kFusedKernels[i]; }
const void* resident_kernel_ptr(bool jax, bool selfhalo) {
    if (jax) return selfhalo ? reinterpret_cast<const void*>(&mcalf_resident_kernel<true, true>)
                             : reinterpret_cast<const void*>(&mcalf_resident_kernel<true, false>);
    return selfhalo ? reinterpret_cast<const void*>(&mcalf_resident_kernel<false, true>)
                    : reinterpret_cast<const void*>(&mcalf_resident_kernel<false, false>);
}
const void* sample_kernel_ptr(bool jax) {
    return jax ? reinterpret_cast<const void*>(&mcalf_sample_kernel<true>) : reinterpret_cast<const void*>(&mcalf_sample_kernel<false>);
}
const void* finalize_kernel_ptr() { return reinterpret_cast<const void*>(&mcalf_finalize_kernel); }
const void* hjert_kernel_ptr() { return reinterpret_cast<const void*>(&mcalf_hjert_kernel); }
const void* scale_cube_kernel_ptr() { return reinterpret_cast<const void*>(&mcalf_scale_cube_kernel); }
const void* xcd_probe_kernel_ptr() { return reinterpret_cast<const void*>(&mcalf_xcd_probe_kernel); }
const void* wide_taps_kernel_ptr(bool jax) {
    return jax ? reinterpret_cast<const void*>(&mcalf_wide_taps_kernel<true>) : reinterpret_cast<const void*>(&mcalf_wide_taps_kernel<false>);
}
const void* wide_conv_kernel_ptr(bool jax) {
    return jax ? reinterpret_cast<const void*>(&mcalf_wide_conv_kernel<true>) : reinterpret_cast<const void*>(&mcalf_wide_conv_kernel<false>);
}
const void* wide_rows_kernel_ptr() { return reinterpret_cast<const void*>(&mcalf_wide_rows_kernel); }

}  // namespace mcalf
