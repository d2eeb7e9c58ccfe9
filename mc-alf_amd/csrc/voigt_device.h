// Device Voigt-Hjerting function H(x, y) = Re w(x + i y) for gfx950 (float64).
//
// Replaces scipy.special.wofz at the reference call site hires_fitter.py:365 and the
// float32 Algorithm-916 `hjert` of voigt_jax.py:121-127.  Not a translation of either:
// the damping parameter y = gamma/(4 pi dnu) is constant per (component, line) and
// tiny (1e-6..1e-2 for every resonance line), so H is expanded in y about the real axis,
//
//   H(x, y) = exp(y^2 - x^2) cos(2 x y)  -  sum_m (-1)^m y^(2m+1) L^(2m+1)(x) / (2m+1)!
//
// with L(x) = Im w(x) = 2 Dawson(x)/sqrt(pi) and L' = 2/sqrt(pi) - 2 x L,
// L^(n+1) = -2 n L^(n-1) - 2 x L^(n)  (from w' = -2 z w + 2i/sqrt(pi)).
//
//   core  |x| <  8 : L and K1 = -L' from piecewise degree-11 tables (LDS), exp(-x^2) direct
//   wing  |x| >= 8 : exp(-x^2) is gone; y K1 + y^3 K3 + y^5 K5 as polynomials in t = 1/x^2
//   general path   : y > 2^-8 (never for physical lines): trapezoid sum with pole
//                    correction (Matta-Reichel / Zaghloul-Ali form) for |z| < 8, Laplace
//                    asymptotic series beyond.
//
// Accuracy (tests/test_gpu_voigt.py): <= 2e-15 relative against 40-digit mpmath on the
// fast path, <= 1e-14 on the general path; scipy.special.wofz itself is ~2e-14.
#pragma once
#include <hip/hip_runtime.h>

#include "voigt_tables.h"

namespace mcalf {

constexpr double kInvSqrtPi = 0.56418958354775628695;   // 1/sqrt(pi)
constexpr double kTwoInvSqrtPi = 1.12837916709551257390;  // 2/sqrt(pi)
constexpr double kYFastMax = 0.00390625;               // 2^-8: upper y of the fast path
constexpr double kX2Core = VT_XCORE * VT_XCORE;         // 64

// ---- wing: returns t * [M1(t) - q (M3(t) - q M5(t))],  q = y^2 t,  t = 1/x^2 ------------
// H = (y / sqrt(pi)) * (that).  x2 >= 64.
__device__ __forceinline__ double hjert_wing_scaled(double x2, double y2) {
    const double t = 1.0 / x2;
    const double q = y2 * t;
    double m1 = VT_M1_HOST[VT_M1DEG];
#pragma unroll
    for (int k = VT_M1DEG - 1; k >= 0; --k) m1 = fma(m1, t, VT_M1_HOST[k]);
    double m3 = VT_M3_HOST[VT_M3DEG];
#pragma unroll
    for (int k = VT_M3DEG - 1; k >= 0; --k) m3 = fma(m3, t, VT_M3_HOST[k]);
    double m5 = VT_M5_HOST[VT_M5DEG];
#pragma unroll
    for (int k = VT_M5DEG - 1; k >= 0; --k) m5 = fma(m5, t, VT_M5_HOST[k]);
    const double inner = fma(-q, fma(-q, m5, m3), m1);
    return t * inner;
}

// ---- core: |x| < 8, 0 <= y <= 2^-8.  tabL / tabK1 point at the [32][12] tables ---------
__device__ __forceinline__ double hjert_core(double x, double x2, double y, double y2, double ey2,
                                             const double* __restrict__ tabL,
                                             const double* __restrict__ tabK1) {
    int j = (int)(x * 4.0);
    j = j > (VT_NINT - 1) ? (VT_NINT - 1) : j;
    const double s = fma(x, 8.0, -(double)(2 * j + 1));
    const double* cL = tabL + j * VT_LSTRIDE;
    const double* cK = tabK1 + j * VT_LSTRIDE;
    double L = cL[VT_LDEG];
    double K1 = cK[VT_LDEG];
#pragma unroll
    for (int k = VT_LDEG - 1; k >= 0; --k) {
        L = fma(L, s, cL[k]);
        K1 = fma(K1, s, cK[k]);
    }
    const double G = exp(-x2);
    // derivatives of L: L1 = -K1
    const double m2x = -2.0 * x;
    const double L1 = -K1;
    const double L2 = fma(m2x, L1, -2.0 * L);
    const double L3 = fma(m2x, L2, -4.0 * L1);
    const double L4 = fma(m2x, L3, -6.0 * L2);
    const double L5 = fma(m2x, L4, -8.0 * L3);
    // odd part:  y K1 + (y^3/6) L3 - (y^5/120) L5
    const double odd = y * fma(y2, fma(-y2 * (1.0 / 120.0), L5, L3 * (1.0 / 6.0)), K1);
    // even part: exp(y^2 - x^2) cos(2 x y),  2xy <= 1/16
    const double q = 2.0 * x * y;
    const double q2 = q * q;
    const double cs = fma(q2, fma(q2, fma(q2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    return fma(G * ey2, cs, odd);
}

// ---- general path: any y >= 0 (used when y > 2^-8) --------------------------------------
__device__ __noinline__ double hjert_general(double x, double y) {
    const double x2 = x * x, y2 = y * y;
    if (x2 + y2 >= 64.0) {
        // Laplace asymptotic series  w(z) ~ (i/sqrt(pi)) (1/z) sum_k (2k-1)!!/(2 z^2)^k ,
        // complex Horner in s = 1/z^2, 14 terms (3e-15 at |z| = 8).
        const double r2 = x2 + y2;
        const double ir2 = 1.0 / r2;
        const double zr = x * ir2, zi = -y * ir2;                 // 1/z
        const double sr = zr * zr - zi * zi, si = 2.0 * zr * zi;  // 1/z^2
        double ck = 1.0;
        double c[15];
        c[0] = 1.0;
#pragma unroll
        for (int k = 1; k <= 14; ++k) { ck *= (2.0 * k - 1.0) * 0.5; c[k] = ck; }
        double ar = c[14], ai = 0.0;
#pragma unroll
        for (int k = 13; k >= 0; --k) {
            const double nr = fma(ar, sr, -ai * si) + c[k];
            const double ni = fma(ar, si, ai * sr);
            ar = nr; ai = ni;
        }
        // w = (i/sqrt(pi)) (1/z) S  ->  Re w = -(1/sqrt(pi)) Im[(1/z) S]
        return -kInvSqrtPi * fma(zr, ai, zi * ar);
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

// Full H(x, y) for arbitrary inputs (diagnostic entry + reference for the in-kernel dispatch).
__device__ inline double hjert(double x, double y, const double* tabL, const double* tabK1) {
    x = fabs(x);
    if (!(y <= kYFastMax)) return hjert_general(x, y);
    const double x2 = x * x, y2 = y * y;
    if (x2 >= kX2Core) {
        double H = y * kInvSqrtPi * hjert_wing_scaled(x2, y2);
        if (x2 < 745.0) H += exp(-x2);   // only matters for y < 2e-10
        return H;
    }
    return hjert_core(x, x2, y, y2, exp(y2), tabL, tabK1);
}

}  // namespace mcalf
