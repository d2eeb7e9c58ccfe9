#!/usr/bin/env python3
"""Round-5 stall experiment: the far-zone Voigt coefficients of a line on the SCALAR path.

Patches a COPY of the device sources (build/far_scalar_src) and builds build/abl/far_scalar.so, to be A/B-ed against the
product library through MCALF_HIP_LIB.  What changes:

  * a (component, line) record grows from 8 to 16 doubles: behind [A, B, x2c, y, K, Kyt, Kgen, uthr] the set-up code
    (setup_sample -> build_line_record) stores the line's seven zone-F coefficients, folded with the line's y and scaled
    by Kyt with exactly the arithmetic of the component loop's fold (same bits);
  * in the batch instantiations of the fused kernel (not the one-launch variant, not the streaming launch) the component
    loop no longer reads a line's ten wave-uniform doubles [cF0..6, uthr, A, B] from LDS (5 x ds_read_b128 per line and
    wave, held in 20 VGPRs): it fetches them with scalar loads from the record in HBM (s_load_dwordx16 + x4 through the
    constant address space), ONE LINE AHEAD, and uses them as SGPR operands of the v_fma_f64 chain;
  * the other instantiations read the same ten doubles from the record's copy in LDS.

Every patch names the exact source text it hooks on and fails loudly when that text has changed."""
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "mc-alf_amd", "csrc")
work = os.path.join(root, "build", "far_scalar_src")
out = os.path.join(root, "build", "abl", "far_scalar.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
sys.path.insert(0, root)
import importlib  # noqa: E402
bld = importlib.import_module("mc-alf_amd.build")
bld.copy_sources(work)


def patch(name, pairs):
    p = os.path.join(work, name)
    s = open(p).read()
    for a, b in pairs:
        if s.count(a) != 1:
            sys.exit("make_far_scalar_build: %s: expected 1 occurrence, found %d, of:\n%s" % (name, s.count(a), a))
        s = s.replace(a, b)
    open(p, "w").write(s)


patch("kernel_args.h", [
    ("constexpr int kRecStride = 8;           // doubles per (component,line) record in LDS",
     "constexpr int kRecStride = 16;          // doubles per (component,line) record: [A, B, x2c, y, K, Kyt, Kgen, uthr, cF0..cF6, pad]\n"
     "constexpr int kRecCF = 8;               // the line's folded zone-F coefficients, right behind uthr"),
])
patch("kernels.hip", [
    # ---- set-up: fold the far-zone coefficients into the record ------------------------------------------------
    ("""__device__ inline void build_line_record(double* rec, double logN, double z, double b_kms, const LineDev& ln,
                                         double dnu_seg) {""",
     """typedef __attribute__((address_space(4))) const double cdouble;     // (uniform loads through it are scalar loads)
__device__ inline void build_line_record(double* rec, double logN, double z, double b_kms, const LineDev& ln,
                                         double dnu_seg, const double* tabs) {"""),
    ("""        rec[0] = 0.0; rec[1] = -1e6; rec[2] = 36.0; rec[3] = 0.0; rec[4] = 0.0; rec[5] = 0.0; rec[6] = 0.0;
        rec[7] = 0.0;
    }
}
""", """        rec[0] = 0.0; rec[1] = -1e6; rec[2] = 36.0; rec[3] = 0.0; rec[4] = 0.0; rec[5] = 0.0; rec[6] = 0.0;
        rec[7] = 0.0;
    }
    // the line's zone-F coefficients, folded with y and scaled by Kyt: the component loop's own fold (fc = T6; fc = fma(fc,
    // y, Tn) for n = 5 .. 0; fc * Kyt), once per (live point, line) instead of once per workgroup and barrier
    {
        cdouble* T = (cdouble*)tabs;
        const double y = rec[3], sc = rec[5];
#pragma unroll
        for (int k = 0; k <= VT_FDEG; ++k) {
            double c = T[(VT_NY - 1) * VT_NTOT + VT_ZF_OFF + k];
#pragma unroll
            for (int nn = VT_NY - 2; nn >= 0; --nn) c = fma(c, y, T[nn * VT_NTOT + VT_ZF_OFF + k]);
            rec[kRecCF + k] = c * sc;
        }
        rec[kRecStride - 1] = 0.0;
    }
}
"""),
    ("        build_line_record(rec, logN, z, b, *ln, a.dnu_seg);", "        build_line_record(rec, logN, z, b, *ln, a.dnu_seg, a.tabs);"),
    # ---- eval_line: the ten line constants arrive as arguments -----------------------------------------------------
    ("""__device__ __forceinline__ void eval_line(const double* __restrict__ tab,
                                          const double (&nu)[kPpt], double (&tau)[kPpt], double nuNode,
                                          double& farNode, unsigned long long segOk) {
    const double A = tab[kLineLds + 1], B = tab[kLineLds + 2];
    double cF[VT_FDEG + 1];
#pragma unroll
    for (int k = 0; k <= VT_FDEG; ++k) cF[k] = tab[kZFLds + k];
""", """// `lc` = the line's wave-uniform constants [A, B, uthr, cF0 .. cF6]
__device__ __forceinline__ void eval_line(const double* __restrict__ tab, const double (&lc)[3 + VT_FDEG + 1],
                                          const double (&nu)[kPpt], double (&tau)[kPpt], double nuNode,
                                          double& farNode, unsigned long long segOk) {
    const double A = lc[0], B = lc[1];
    double cF[VT_FDEG + 1];
#pragma unroll
    for (int k = 0; k <= VT_FDEG; ++k) cF[k] = lc[3 + k];
"""),
    ("        const double uthr = tab[kLineLds];\n", "        const double uthr = lc[2];\n"),
    # ---- the component loop --------------------------------------------------------------------------------------------
    ("""        int buf = 0;
        const int ncl_run = ncl;
""", """        int buf = 0;
        const int ncl_run = ncl;
        // The ten wave-uniform constants of a line.  Batch instantiations: scalar loads from the live point's records in HBM
        // (written by the set-up kernel of the same stream), one line ahead of their use; the others: the records' copy in LDS.
        constexpr bool kScalarFar = !kInline && !kStream;
        constexpr int kLc = 3 + VT_FDEG + 1;
        auto load_lc = [&](int cl, double (&lc)[kLc]) {
            const int c = min(cl, max(ncl_run - 1, 0));
            if constexpr (kScalarFar) {
                const int su = __builtin_amdgcn_readfirstlane(s);
                cdouble* gr = (cdouble*)(a.recs + (size_t)su * recTotal + (size_t)c * kRecStride);
                lc[0] = gr[0]; lc[1] = gr[1];
#pragma unroll
                for (int k = 0; k < kLc - 2; ++k) lc[2 + k] = gr[7 + k];            // uthr, cF0 .. cF6: doubles 7 .. 14
            } else {
                const double* lr = sRec + c * kRecStride;
                lc[0] = lr[0]; lc[1] = lr[1];
#pragma unroll
                for (int k = 0; k < kLc - 2; ++k) lc[2 + k] = lr[7 + k];
            }
        };
        double lcCur[kLc], lcNext[kLc];
        if constexpr (kScalarFar) load_lc(0, lcCur);
"""),
    ("""#pragma unroll 1
            for (int l = 0; l < lmax; ++l) eval_line(tabs + l * kTabPad, nu, tau, nuNode, farNode, segOk);
""", """#pragma unroll 1
            for (int l = 0; l < lmax; ++l) {
                if constexpr (kScalarFar) {
                    load_lc(cl0 + l + 1, lcNext);                    // (the next line's, also across the barrier of the next group)
                } else {
                    load_lc(cl0 + l, lcCur);
                }
                eval_line(tabs + l * kTabPad, lcCur, nu, tau, nuNode, farNode, segOk);
                if constexpr (kScalarFar) {
#pragma unroll
                    for (int k = 0; k < kLc; ++k) lcCur[k] = lcNext[k];
                }
            }
"""),
])

log = bld.build_tree(work, out, stamp="far_scalar_experiment", report=True)
for row in bld.resource_table(log):
    print("%-34s VGPR %3d scratch %3d occupancy %d sgpr-spill %3d vgpr-spill %3d" % ((row[0][-34:],) + row[1:]))
print(out)
