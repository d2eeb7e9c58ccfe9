#!/usr/bin/env python3
"""Generate `mc-alf_amd/csrc/voigt_tables.h`: the universal coefficient table of the device
Voigt-Hjerting function H(x, y) = Re w(x + iy), small-damping fast path.

Everything is computed from the definition of the Faddeeva function with mpmath at 60
digits; no coefficient is taken from any library.  DESIGN.md ("Voigt function") has the
derivation; in short, for 0 <= y <= 2^-8

    H(x, y) = sum_{n=0..6} y^n h_n(x) + O(y^7),     h_n(x) = Re[i^n w^(n)(x)] / n!

(w^(n) from w' = -2 z w + 2i/sqrt(pi)).  y is constant per (component, line), so the kernel
FOLDS y into the coefficients once per line:  c[idx] = scale * sum_n y^n T[n][idx],  and the
per-pixel work is a single Horner evaluation.

Layout of T[n][idx], n = 0..6, idx in [0, VT_NTOT):
  core   idx = j*12 + k, j < 32 : |x| in [j/4, (j+1)/4), polynomial in s = 8|x| - (2j+1), coefficient k
         T[n] = Chebyshev-interpolant monomial coefficients of h_n
  zone1  idx = 384 + k, k < 11  : 6 <= |x| < 8, variable s = VT_Z1_A * t + VT_Z1_B in [-1,1], t = 1/x^2
  zone0  idx = 395 + k, k < 11  : 8 <= |x| < 16, variable t
  zoneF  idx = 406 + k, k < 7   : |x| >= 16, variable t, degree 6
         exp(-x^2) is dropped (the kernel only enters these zones where K exp(-x^2) < 2e-17):
         H = (y t/sqrt(pi)) [Q1 - y^2 Q3 + y^4 Q5](t),  Q1 = sqrt(pi) x^2 h_1,
         Q3 = -sqrt(pi) x^2 h_3,  Q5 = sqrt(pi) x^2 h_5;   T[0] = Q1, T[2] = -Q3, T[4] = Q5, odd rows 0.
"""
import os
import sys

import mpmath as mp

mp.mp.dps = 60
SQPI = mp.sqrt(mp.pi)

XCORE = 8
NINT = 32
CDEG = 11                      # core polynomial degree
WDEG = 10                      # wing polynomial degree
NY = 7                         # powers of y kept (n = 0..6)
XMID = 6                       # zone1 covers [XMID, XCORE)
NCORE = NINT * (CDEG + 1)
Z1_OFF = NCORE
Z0_OFF = NCORE + (WDEG + 1)
FDEG = 6                       # far-wing polynomial degree (|x| >= XFAR)
XFAR = 16
ZF_OFF = NCORE + 2 * (WDEG + 1)
NTOT = ZF_OFF + (FDEG + 1)


def w_derivs(x, nmax):
    """w^(n)(x) for real x, n = 0..nmax."""
    x = mp.mpf(x)
    w0 = mp.exp(-x * x) * mp.erfc(-1j * x)
    w = [w0, -2 * x * w0 + 2j / SQPI]
    for n in range(1, nmax):
        w.append(-2 * x * w[n] - 2 * n * w[n - 1])
    return w


def h_all(x):
    w = w_derivs(x, NY - 1)
    return [mp.re((1j) ** n * w[n]) / mp.factorial(n) for n in range(NY)]


def cheb_coefs(f, a, b, deg):
    c = (mp.mpf(a) + b) / 2
    h = (mp.mpf(b) - a) / 2
    n = deg + 1
    nodes = [mp.cos(mp.pi * (mp.mpf(i) + mp.mpf(1) / 2) / n) for i in range(n)]
    vals = [f(c + h * s) for s in nodes]
    out = []
    for j in range(n):
        acc = sum(vals[i] * mp.cos(mp.pi * j * (mp.mpf(i) + mp.mpf(1) / 2) / n) for i in range(n))
        acc = acc * 2 / n
        if j == 0:
            acc /= 2
        out.append(acc)
    return out


def cheb_to_mono(cc):
    n = len(cc)
    T = [[mp.mpf(0)] * n for _ in range(n)]
    T[0][0] = mp.mpf(1)
    if n > 1:
        T[1][1] = mp.mpf(1)
    for k in range(2, n):
        for j in range(n):
            T[k][j] = (2 * T[k - 1][j - 1] if j > 0 else 0) - T[k - 2][j]
    mono = [mp.mpf(0)] * n
    for k in range(n):
        for j in range(n):
            mono[j] += cc[k] * T[k][j]
    return mono


def mono_shift(ms, c, h):
    """coefficients in s = (t - c)/h  ->  coefficients in t."""
    n = len(ms)
    out = [mp.mpf(0)] * n
    for j in range(n):
        for i in range(j + 1):
            out[i] += ms[j] * mp.binomial(j, i) * (-c) ** (j - i) / h ** j
    return out


def q_of(t, which):
    """Q1, Q3, Q5 at t = 1/x^2 (t = 0 -> asymptotic limits 1, 0, 0)."""
    if t == 0:
        return mp.mpf(1) if which == 1 else mp.mpf(0)
    x = 1 / mp.sqrt(t)
    h = h_all(x)
    if which == 1:
        return SQPI * x * x * h[1]
    if which == 3:
        return -SQPI * x * x * h[3]
    return SQPI * x * x * h[5]


def build():
    T = [[mp.mpf(0)] * NTOT for _ in range(NY)]
    w = mp.mpf(XCORE) / NINT
    for j in range(NINT):
        a, b = j * w, (j + 1) * w
        for n in range(NY):
            mono = cheb_to_mono(cheb_coefs(lambda x, n=n: h_all(x)[n], a, b, CDEG))
            for k in range(CDEG + 1):
                T[n][j * (CDEG + 1) + k] = mono[k]
    # zone1: t in [1/64, 1/36], centred variable
    t_lo, t_hi = mp.mpf(1) / (XCORE * XCORE), mp.mpf(1) / (XMID * XMID)
    tc, th = (t_lo + t_hi) / 2, (t_hi - t_lo) / 2
    for which, row, sign in ((1, 0, 1), (3, 2, -1), (5, 4, 1)):
        mono = cheb_to_mono(cheb_coefs(lambda t, wh=which: q_of(t, wh), t_lo, t_hi, WDEG))
        for k in range(WDEG + 1):
            T[row][Z1_OFF + k] = sign * mono[k]
        mono0 = mono_shift(cheb_to_mono(cheb_coefs(lambda t, wh=which: q_of(t, wh), 0, t_lo, WDEG)), t_lo / 2, t_lo / 2)
        for k in range(WDEG + 1):
            T[row][Z0_OFF + k] = sign * mono0[k]
        t_far = mp.mpf(1) / (XFAR * XFAR)
        monof = mono_shift(cheb_to_mono(cheb_coefs(lambda t, wh=which: q_of(t, wh), 0, t_far, FDEG)), t_far / 2, t_far / 2)
        for k in range(FDEG + 1):
            T[row][ZF_OFF + k] = sign * monof[k]
    return T, (1 / th, -tc / th)


INTERP_NODES = [0, 3, 12, 24, 39, 51, 60, 63]     # near Chebyshev-Lobatto positions in a 64-pixel segment


def interp_weights():
    """W[p][k]: Lagrange basis of node k at in-segment position p (exact rationals -> double)."""
    W = []
    for p in range(64):
        row = []
        for k, xk in enumerate(INTERP_NODES):
            num = mp.mpf(1)
            for m, xm in enumerate(INTERP_NODES):
                if m != k:
                    num *= mp.mpf(p - xm) / mp.mpf(xk - xm)
            row.append(num)
        W.append(row)
    return W


def main(out_path):
    T, (z1a, z1b) = build()
    L = []
    L.append("// GENERATED by tools/gen_voigt_tables.py -- do not edit.")
    L.append("// Universal table T[n][idx] of the device Voigt-Hjerting function (see DESIGN.md).")
    L.append("#pragma once")
    L.append(f"#define VT_XCORE {float(XCORE)!r}")
    L.append(f"#define VT_XMID {float(XMID)!r}")
    L.append(f"#define VT_NINT {NINT}")
    L.append(f"#define VT_CDEG {CDEG}")
    L.append(f"#define VT_CSTRIDE {CDEG + 1}")
    L.append(f"#define VT_WDEG {WDEG}")
    L.append(f"#define VT_NY {NY}")
    L.append(f"#define VT_NCORE {NCORE}")
    L.append(f"#define VT_Z1_OFF {Z1_OFF}")
    L.append(f"#define VT_Z0_OFF {Z0_OFF}")
    L.append(f"#define VT_ZF_OFF {ZF_OFF}")
    L.append(f"#define VT_FDEG {FDEG}")
    L.append(f"#define VT_XFAR {float(XFAR)!r}")
    L.append(f"#define VT_NTOT {NTOT}")
    L.append(f"#define VT_Z1_A {float(z1a)!r}")
    L.append(f"#define VT_Z1_B {float(z1b)!r}")
    L.append(f"static const double VT_T_HOST[{NY} * {NTOT}] = {{")
    for n in range(NY):
        L.append(f"  // n = {n}")
        row = T[n]
        for i in range(0, NTOT, 6):
            L.append("  " + ", ".join(repr(float(v)) for v in row[i:i + 6]) + ",")
    L.append("};")
    L.append("// Far-wing interpolation over a 64-pixel segment: 8 nodes (pixel positions) and the Lagrange")
    L.append("// weights W[p][k] in pixel-index space (DESIGN.md, 'far-wing interpolation').")
    L.append("#define VT_INODES 8")
    L.append("static const int VT_INTERP_NODES[8] = {" + ", ".join(str(v) for v in INTERP_NODES) + "};")
    L.append("static const double VT_INTERP_W_HOST[64 * 8] = {")
    for row in interp_weights():
        L.append("  " + ", ".join(repr(float(v)) for v in row) + ",")
    L.append("};")
    with open(out_path, "w") as fh:
        fh.write("\n".join(L) + "\n")
    print("wrote", out_path)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    default = os.path.join(here, "..", "mc-alf_amd", "csrc", "voigt_tables.h")
    main(sys.argv[1] if len(sys.argv) > 1 else default)
