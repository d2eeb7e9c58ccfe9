// Device Voigt-Hjerting function H(x, y) = Re w(x + i y) for gfx950 (float64).
//
// Replaces scipy.special.wofz at the reference call site hires_fitter.py:365 and the
// float32 Algorithm-916 `hjert` of voigt_jax.py:121-127.  Not a translation of either.
//
// The damping parameter y = gamma/(4 pi dnu) is constant per (component, line) and tiny
// (1e-6..4e-3 for every resonance line), so H is expanded in y about the real axis:
//
//     H(x, y) = sum_{n=0..6} y^n h_n(x),     h_n(x) = Re[i^n w^(n)(x)] / n!
//
// and y is FOLDED into polynomial coefficients once per line (fold_coef below): the
// universal table T[n][idx] (tools/gen_voigt_tables.py, from 60-digit mpmath) holds, for each
// of 32 intervals of |x| < 8 and for two wing zones in t = 1/x^2, the coefficients of every
// h_n; after folding, one H evaluation is ONE Horner pass:
//
//   core  |x| < x_c      : degree 11 in s = 8|x| - (2j+1),  j = floor(4|x|)   (per-lane LDS gather)
//   zone1 x_c <= |x| < 8 : degree 10 in s = A t + B, times t                  (exp(-x^2) dropped)
//   zone0 8 <= |x| < 16  : degree 10 in t, times t
//   zoneF |x| >= 16      : degree  6 in t, times t   (most pixels of a spectrum)
//
// x_c in [6, 8] is chosen per line so that the dropped K exp(-x_c^2) < 2e-17 in optical depth.
// General path (y > 2^-8, never reached by physical lines): trapezoid sum with pole correction
// for |z| < 8, Laplace asymptotic series beyond.
//
// Accuracy (tests/test_gpu_voigt.py): fast path |dH| <= 3e-17 + 5e-15 H against 40-digit
// mpmath; general path <= 2e-14 relative; scipy.special.wofz itself is ~2e-14.
#pragma once
#include <hip/hip_runtime.h>

#include "voigt_tables.h"

namespace mcalf {

constexpr double kInvSqrtPi = 0.56418958354775628695;   // 1/sqrt(pi)
constexpr double kYFastMax = 0.00390625;                // 2^-8: upper y of the fast path
constexpr double kX2Wing = VT_XCORE * VT_XCORE;         // 64
constexpr double kX2Mid = VT_XMID * VT_XMID;            // 36
constexpr double kX2Far = VT_XFAR * VT_XFAR;            // 256
constexpr double kDropLog = 38.5;                       // -ln(2e-17): K exp(-x^2) < 2e-17  <=>  x^2 > ln K + 38.5

// x_c^2 for a line of optical-depth scale K
__device__ __forceinline__ double core_limit_x2(double K) {
    const double v = (double)__logf((float)K) + kDropLog;   // 1e-6 is plenty for a switch-over point
    return fmin(fmax(v, kX2Mid), kX2Wing);      // NaN -> kX2Mid via fmax/fmin semantics
}

// c[idx] = scale * sum_n y^n T[n][idx]
__device__ __forceinline__ double fold_coef(const double (&Tn)[VT_NY], double y, double scale) {
    double c = Tn[VT_NY - 1];
#pragma unroll
    for (int n = VT_NY - 2; n >= 0; --n) c = fma(c, y, Tn[n]);
    return c * scale;
}

// 1/x for x in [36, 1e300]: v_rcp_f64 (measured 4.5e-8 relative on gfx950) + one Newton step
// -> 2.1e-15 relative (tools/micro/rcp_test.hip); no division fix-up needed in this range.
__device__ __forceinline__ double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}

// d = a * b + c with c held in a scalar register pair.  The compiler otherwise materialises every 64-bit
// polynomial constant with two v_mov_b32 per use (as costly on the vector pipe as the FMA itself); scalar
// moves are free next to a saturated VALU and are shared by the thread's 8 pixels.
__device__ __forceinline__ double fma_sconst(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// exp(-t): the argument reduction and degree-11 polynomial of the ROCm device library's exp (n =
// rint(x log2 e), r = x - n ln2_hi - n ln2_lo, 2^n p(r)), with the constants in scalar registers.
// Same result as exp(-t) to the last bit (tools/micro/exp_test.hip), including exp(-inf) = 0,
// exp(+inf) = inf and NaN.
__device__ __forceinline__ double exp_neg(double t) {
    const double n = __builtin_rint(t * -0x1.71547652b82fep+0);            // x log2(e), x = -t
    double r = fma(n, -0x1.62e42fefa39efp-1, -t);                          // x - n ln2_hi
    r = fma(n, -0x1.abc9e3b39803fp-56, r);                                 //   - n ln2_lo
    double p = fma_sconst(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma_sconst(r, p, 0x1.71dee623fde64p-19);
    p = fma_sconst(r, p, 0x1.a01997c89e6b0p-16);
    p = fma_sconst(r, p, 0x1.a01a014761f6ep-13);
    p = fma_sconst(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma_sconst(r, p, 0x1.1111111122322p-7);
    p = fma_sconst(r, p, 0x1.55555555502a1p-5);
    p = fma_sconst(r, p, 0x1.5555555555511p-3);
    p = fma_sconst(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double v = ldexp(p, (int)n);
    v = (t < -1024.0) ? INFINITY : v;
    return (t > 1075.0) ? 0.0 : v;
}

// ---- general path: y > 2^-8, absurd columns, and y < 0 (a negative Doppler parameter) ------
// Upper half plane, x >= 0.
__device__ __forceinline__ double hjert_upper(double x, double y) {
    const double x2 = x * x, y2 = y * y;
    if (x2 + y2 >= 64.0) {
        // Laplace asymptotic series  w(z) ~ (i/sqrt(pi)) (1/z) sum_k (2k-1)!!/(2 z^2)^k ,
        // complex Horner in s = 1/z^2, 14 terms (3e-15 at |z| = 8).
        const double ir2 = 1.0 / (x2 + y2);
        const double zr = x * ir2, zi = -y * ir2;                 // 1/z
        const double sr = zr * zr - zi * zi, si = 2.0 * zr * zi;  // 1/z^2
        double c[15];
        c[0] = 1.0;
#pragma unroll
        for (int k = 1; k <= 14; ++k) c[k] = c[k - 1] * (2.0 * k - 1.0) * 0.5;
        double ar = c[14], ai = 0.0;
#pragma unroll
        for (int k = 13; k >= 0; --k) {
            const double nr = fma(ar, sr, -ai * si) + c[k];
            const double ni = fma(ar, si, ai * sr);
            ar = nr; ai = ni;
        }
        // w = (i/sqrt(pi)) (1/z) S  ->  Re w = -(1/sqrt(pi)) Im[(1/z) S]
        double H = -kInvSqrtPi * fma(zr, ai, zi * ar);
        if (x2 < 745.0 && y < 1e-3) H += exp(y2 - x2) * cos(2.0 * x * y);   // beyond all orders of the series
        return H;
    }
    // |z| < 8: trapezoid rule (step h = 1/2) on the convolution integral with the pole
    // correction terms; discretisation error exp(-pi^2/h^2) = 7e-18.
    const double h = 0.5;
    const double G = exp(-x2);
    const double xy = x * y;
    double s1, c1;
    sincos(xy, &s1, &c1);
    const double c2 = fma(-2.0 * s1, s1, 1.0);                   // cos(2xy)
    const double T1 = G * erfcx(y) * c2;
    const double T2 = (y > 0.0) ? (2.0 * h) * G * (s1 * s1) / (M_PI * y) : 0.0;
    double S1 = 0.0, S23 = 0.0;
    for (int n = 1; n <= 32; ++n) {
        const double hn = h * n;
        const double d = 1.0 / fma(hn, hn, y2);
        S1 = fma(exp(-hn * hn), d, S1);
        const double ep = hn + x, em = hn - x;
        S23 = fma(exp(-ep * ep) + exp(-em * em), d, S23);
    }
    S1 *= G;
    return T1 + T2 + (2.0 * h * y / M_PI) * fma(-c2, S1, 0.5 * S23);
}

// Any real y (x >= 0).  Lower half plane by reflection, w(z) = 2 exp(-z^2) - w(-z), which is what
// scipy.special.wofz returns there (the reference reaches it with b < 0: a = gamma/(4 pi dnu) < 0).
__device__ __noinline__ double hjert_general(double x, double y) {
    if (!(y < 0.0)) return hjert_upper(x, y);
    const double ay = -y;
    return 2.0 * exp((ay - x) * (x + ay)) * cos(2.0 * x * ay) - hjert_upper(x, ay);
}

// Full H(x, y) with exactly the kernel's arithmetic for a line of optical-depth scale K = 1 (diagnostic
// entries).  T is the [VT_NY][VT_NTOT] table in global memory.  node_form = false: what a directly evaluated
// pixel gets (core table for every |x| < 8); true: what an interpolation node gets (zone 1, exp(-x^2) dropped,
// between x_c and 8).
__device__ inline double hjert_folded(double x, double y, const double* __restrict__ T, bool node_form) {
    x = fabs(x);
    if (!(y <= kYFastMax) || !(y >= 0.0)) return hjert_general(x, y);
    const double x2 = x * x;
    const double x2c = node_form ? core_limit_x2(1.0) : kX2Wing;
    double Tn[VT_NY];
    if (x2 >= x2c) {
        const double t = fast_rcp(x2);
        const bool wing = x2 >= kX2Wing;
        const bool far = x2 >= kX2Far;
        const int off = far ? VT_ZF_OFF : (wing ? VT_Z0_OFF : VT_Z1_OFF);
        const double s = wing ? t : fma(t, VT_Z1_A, VT_Z1_B);
        double P = 0.0;
        for (int k = far ? VT_FDEG : VT_WDEG; k >= 0; --k) {
            for (int n = 0; n < VT_NY; ++n) Tn[n] = T[n * VT_NTOT + off + k];
            P = fma(P, s, fold_coef(Tn, y, y * kInvSqrtPi));
        }
        return t * P;
    }
    int j = (int)(x * 4.0);
    j = min(max(j, 0), VT_NINT - 1);
    const double s = fma(x, 8.0, -(double)(2 * j + 1));
    double P = 0.0;
    for (int k = VT_CDEG; k >= 0; --k) {
        for (int n = 0; n < VT_NY; ++n) Tn[n] = T[n * VT_NTOT + j * VT_CSTRIDE + k];
        P = fma(P, s, fold_coef(Tn, y, 1.0));
    }
    return P;
}

}  // namespace mcalf
