#!/usr/bin/env python3
"""Round-6 stall experiment: raise the instruction-level parallelism of the NODE PASS of the
component loop instead of moving its operands.  Patches COPIES of the device sources and builds

    build/abl/lib_pair.so     two lines per trip of the line loop: both node passes written step by step ACROSS the lines
                              (two independent rcp -> Horner chains issue interleaved); the directly evaluated pixels stay
                              per line, ONE copy of that body (a 2-trip loop), its line constants re-read from LDS
    build/abl/lib_estrin.so   one line per trip as in the product, the far-zone node polynomial (degree 6 in t) evaluated
                              by Estrin's scheme (depth 4 instead of 6, three independent FMAs at its widest) -- NOT
                              bit-identical to the product (another rounding order, same 1e-16 accuracy)
    build/abl/lib_product.so  the product sources, built the same way

to be A/B-ed through MCALF_HIP_LIB (tools/abl_bench.sh, tools/explore/ab_bits.py).  Every patch names the exact source
text it hooks on and fails loudly when that text has changed.   python tools/explore/make_ilp_builds.py [report]"""
import importlib
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
bld = importlib.import_module("mc-alf_amd.build")
outdir = os.path.join(root, "build", "abl")
os.makedirs(outdir, exist_ok=True)


def patched(name, pairs):
    work = os.path.join(root, "build", "ilp_%s_src" % name)
    bld.copy_sources(work)
    p = os.path.join(work, "kernels.hip")
    s = open(p).read()
    for a, b in pairs:
        if s.count(a) != 1:
            sys.exit("make_ilp_builds (%s): expected 1 occurrence, found %d, of:\n%s" % (name, s.count(a), a))
        s = s.replace(a, b)
    open(p, "w").write(s)
    log = bld.build_tree(work, os.path.join(outdir, "lib_%s.so" % name), stamp="ilp-" + name, report=True)
    import subprocess
    rows = bld.resource_table(log)
    print(name)
    for r in rows:
        nm = subprocess.run(["c++filt", r[0]], capture_output=True, text=True).stdout.strip().replace("void mcalf::", "").replace("(mcalf::KArgs)", "")
        if nm.startswith("mcalf_fused_kernel"):
            print("   %-52s VGPRs %3d scratch %3d occupancy %d SGPR spills %3d VGPR spills %2d" % (nm[:52], *r[1:]))


PAIR_FN = r'''
// ---- experiment: two lines per trip.  Node passes of both lines written across the lines; direct pixels per line. ----
__device__ __forceinline__ unsigned long long node_mask(double un, double uthr, unsigned long long segOk) {
    unsigned long long mp = __builtin_amdgcn_ballot_w64(un >= uthr);
    unsigned long long mn = __builtin_amdgcn_ballot_w64(un <= -uthr);
    mp &= mp >> 7;
    mn &= mn >> 7;
    return uniform64((mp | mn) & segOk);
}
__device__ __forceinline__ bool lanes_of(unsigned long long done) {
    const unsigned long long lanes = ((unsigned long long)((unsigned)(done >> 32) * 0xFFu) << 32) | ((unsigned)done * 0xFFu);
    return __builtin_amdgcn_inverse_ballot_w64(lanes);
}
// the product's node polynomial of ONE line (any zone), as in eval_line
__device__ __forceinline__ void node_poly_one(const double* __restrict__ tab, bool mine, double x2n, double t, double& farNode) {
    double P;
    if (!mine || x2n >= kX2Far) {
        P = tab[kZFLds + VT_FDEG];
#pragma unroll
        for (int k = VT_FDEG - 1; k >= 0; --k) P = fma(P, t, tab[kZFLds + k]);
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
__device__ __forceinline__ void direct_pixels(const double* __restrict__ tab, unsigned long long done, const double (&nu)[kPpt], double (&tau)[kPpt]) {
    const unsigned doneLo = (unsigned)done, doneHi = (unsigned)(done >> 32);
    if (doneLo == 0x01010101u && doneHi == 0x01010101u) return;
    const double A = tab[kLineLds + 1], B = tab[kLineLds + 2];
#pragma unroll
    for (int h = 0; h < kPpt / 4; ++h) {
    const unsigned dh = h ? doneHi : doneLo;
    if (dh == 0x01010101u) continue;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;
        if (__builtin_expect(((dh >> (8 * jj)) & 1u) != 0u, 1)) continue;
        const double u = fma(nu[j], A, -B);
        const double x2 = u * u;
        double t, P;
        if (x2 >= kX2Far) {
            t = fast_rcp(x2);
            P = tab[kZFLds + VT_FDEG];
#pragma unroll
            for (int k = VT_FDEG - 1; k >= 0; --k) P = fma(P, t, tab[kZFLds + k]);
        } else if (x2 >= kX2Wing) {
            t = fast_rcp(x2);
            const double* cw = tab + kZ0Lds;
            P = cw[VT_WDEG];
#pragma unroll
            for (int k = VT_WDEG - 1; k >= 0; --k) P = fma(P, t, cw[k]);
        } else {
            const double x4 = fabs(u) * 4.0;
            const int jx = (int)x4;
            const double sv = fma(x4, 2.0, -(double)(2 * jx + 1));
            const double* cc = tab + jx * VT_CSTRIDE;
            P = cc[VT_CDEG];
#pragma unroll
            for (int k = VT_CDEG - 1; k >= 0; --k) P = fma(P, sv, cc[k]);
            t = 1.0;
        }
        fmac_inplace(tau[j], t, P);
    }
    }
}
// lines at tab0 and (when `two`) tab0 + kTabPad
__device__ __forceinline__ void eval_line_pair(const double* __restrict__ tab0, bool two, const double (&nu)[kPpt], double (&tau)[kPpt],
                                               double nuNode, double& farNode, unsigned long long segOk) {
    const double* tab1 = tab0 + (two ? kTabPad : 0);
    const double un0 = fma(nuNode, tab0[kLineLds + 1], -tab0[kLineLds + 2]);
    const double un1 = fma(nuNode, tab1[kLineLds + 1], -tab1[kLineLds + 2]);
    const double x2n0 = un0 * un0, x2n1 = un1 * un1;
    const double t0 = fast_rcp(fmax(x2n0, 4.0)), t1 = fast_rcp(fmax(x2n1, 4.0));
    const unsigned long long done0 = node_mask(un0, tab0[kLineLds], segOk);
    const unsigned long long done1 = two ? node_mask(un1, tab1[kLineLds], segOk) : 0ull;
    const bool mine0 = lanes_of(done0), mine1 = lanes_of(done1);
    const bool far0 = !mine0 || x2n0 >= kX2Far, far1 = !mine1 || x2n1 >= kX2Far;
    if (done0 != 0 && done1 != 0 && __builtin_amdgcn_ballot_w64(far0 && far1) == ~0ull) {     // the usual case: two far chains
        double P0 = tab0[kZFLds + VT_FDEG], P1 = tab1[kZFLds + VT_FDEG];
#pragma unroll
        for (int k = VT_FDEG - 1; k >= 0; --k) {
            P0 = fma(P0, t0, tab0[kZFLds + k]);
            P1 = fma(P1, t1, tab1[kZFLds + k]);
        }
        fmac_inplace(farNode, mine0 ? t0 : 0.0, P0);
        fmac_inplace(farNode, mine1 ? t1 : 0.0, P1);
    } else {
        if (done0 != 0) node_poly_one(tab0, mine0, x2n0, t0, farNode);
        if (done1 != 0) node_poly_one(tab1, mine1, x2n1, t1, farNode);
    }
    const int nl = two ? 2 : 1;
#pragma unroll 1
    for (int i = 0; i < nl; ++i) direct_pixels(i ? tab1 : tab0, i ? done1 : done0, nu, tau);
}
'''

PAIR = [
    ("// theta[i] of one sample: either the row element itself or, with unit-cube input, cube*ptp + min with the",
     PAIR_FN + "\n// theta[i] of one sample: either the row element itself or, with unit-cube input, cube*ptp + min with the"),
    ("""#pragma unroll 1
            for (int l = 0; l < lmax; ++l) eval_line(tabs + l * kTabPad, nu, tau, nuNode, farNode, segOk);""",
     """#pragma unroll 1
            for (int l = 0; l < lmax; l += 2) eval_line_pair(tabs + l * kTabPad, l + 1 < lmax, nu, tau, nuNode, farNode, segOk);"""),
]

ESTRIN = [
    ("""            if (!mine || x2n >= kX2Far) {                         // (lanes of directly evaluated segments never need a wing zone)
                P = cF[VT_FDEG];
#pragma unroll
                for (int k = VT_FDEG - 1; k >= 0; --k) P = fma(P, t, cF[k]);
            } else {""",
     """            if (!mine || x2n >= kX2Far) {                         // (lanes of directly evaluated segments never need a wing zone)
                static_assert(VT_FDEG == 6, "Estrin's scheme below is written for degree 6");
                const double t2 = t * t;
                const double e01 = fma(cF[1], t, cF[0]), e23 = fma(cF[3], t, cF[2]), e45 = fma(cF[5], t, cF[4]);
                const double t4 = t2 * t2;
                const double lo = fma(e23, t2, e01), hi = fma(cF[6], t2, e45);
                P = fma(hi, t4, lo);
            } else {"""),
]

if __name__ == "__main__":
    patched("product", [])
    patched("pair", PAIR)
    patched("estrin", ESTRIN)
