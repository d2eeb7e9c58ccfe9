#!/usr/bin/env python3
"""Experiment (round 4): the fused kernel with another workgroup geometry -- more waves per SIMD with a smaller register
tile -- as a PATCHED COPY of the product source (build/geom_src), compiled into build/abl/geom_<threads>x<ppt>.so.

    python tools/make_geom_build.py --threads 640 --ppt 7 --waves 5 [--cf-lds]
    MCALF_STREAM=0 MCALF_HIP_LIB=build/abl/geom_640x7.so python bench.py ...

640 threads x 7 pixels: 10 waves per workgroup, two workgroups per CU = 5 waves per SIMD (<= 96 VGPRs);
768 threads x 6 pixels: 12 waves, 6 per SIMD (<= 80 VGPRs).  The tile stays 4096 pixels (64 segments), the threads past
it idle.  --cf-lds: the far-zone coefficients are read from LDS where they are used instead of being held in 14 registers.
Every patch names the exact source text it hooks on and fails loudly when that text has changed."""
import argparse
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mc-alf_amd", "csrc")
work = os.path.join(root, "build", "geom_src")
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=640)
ap.add_argument("--ppt", type=int, default=7)
ap.add_argument("--waves", type=int, default=5, help="waves per SIMD the kernel is compiled for")
ap.add_argument("--cf-lds", action="store_true")
ap.add_argument("--report", action="store_true", help="print the kernels' register usage")
args = ap.parse_args()
os.makedirs(os.path.join(root, "build", "abl"), exist_ok=True)
sys.path.insert(0, root)
import importlib  # noqa: E402
bld = importlib.import_module("mc-alf_amd.build")
bld.copy_sources(work)
FILES = ["kernels.hip", "kernel_args.h", "host_abi.cpp"]
text = {f: open(os.path.join(work, f)).read() for f in FILES}


def rep(a, b, count=1):
    """Replace the hook `a` in whichever source file carries it (exactly `count` occurrences over all of them)."""
    found = sum(text[f].count(a) for f in FILES)
    if found != count:
        sys.exit("make_geom_build: expected %d occurrence(s), found %d, of:\n%s" % (count, found, a))
    for f in FILES:
        text[f] = text[f].replace(a, b)


rep("constexpr int kBlock = 512;", "constexpr int kBlock = %d;" % args.threads)
rep("constexpr int kPpt = 8;  ", "constexpr int kPpt = %d;  " % args.ppt)
rep("constexpr int kExtMax = kBlock * kPpt;", "constexpr int kThreadPix = kBlock * kPpt;\nconstexpr int kExtMax = 4096;")
rep('static_assert(kBlock == 64 * VT_INODES, "one interpolation weight per thread");\n', "")
rep('static_assert(kPpt == 8, "the skip tests of eval_line treat the eight segments of a wave as two halves");\n', "")
rep("constexpr int kMinWaves = 4;  ", "constexpr int kMinWaves = %d;  " % args.waves)
rep("    if (kFarInterp) sWt[(tid0 & 7) * 64 + (tid0 >> 3)] = a.wtab[tid0];", "    if (kFarInterp && tid0 < 64 * VT_INODES) sWt[(tid0 & 7) * 64 + (tid0 >> 3)] = a.wtab[tid0];")
rep("""    for (int h = 0; h < kPpt / 4; ++h) {
    const unsigned dh = h ? doneHi : doneLo;
    if (dh == 0x01010101u) continue;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;""", """    for (int h = 0; h < (kPpt + 3) / 4; ++h) {
    const unsigned dh = h ? doneHi : doneLo;
    const int nj = (kPpt - 4 * h) < 4 ? (kPpt - 4 * h) : 4;
    const unsigned all = nj == 4 ? 0x01010101u : (nj == 3 ? 0x00010101u : (nj == 2 ? 0x00000101u : 0x00000001u));
    if ((dh & all) == all) continue;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (jj >= nj) break;
        const int j = 4 * h + jj;""")
rep("        int e = ext0 + 64 * wv + kBlock * (ln >> 3) + VT_INTERP_NODES[ln & 7];",
    "        int e = ext0 + 64 * wv + kBlock * ((ln >> 3) < kPpt ? (ln >> 3) : 0) + VT_INTERP_NODES[ln & 7];")
rep("            for (int j = 0; j < kPpt; ++j) segOk |= ((L.tileMask >> (wv + 8 * j)) & 1ULL) << (8 * j);",
    "            for (int j = 0; j < kPpt; ++j) segOk |= ((wv + kWaves * j) < 64 ? ((L.tileMask >> (wv + kWaves * j)) & 1ULL) : 0ULL) << (8 * j);")
rep('        static_assert(kPpt % 2 == 0, "pixels are processed in pairs");\n', "")
rep("""            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                double tj = tau[j];""", """            for (int jj = 0; jj < 2; ++jj) {
                const int j = (j0 + jj) < kPpt ? (j0 + jj) : (kPpt - 1);
                double tj = tau[j];""")
rep("""            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                if (selfHalo) {""", """            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                if (j >= kPpt) break;
                if (selfHalo) {""")
rep("            if (r >= a.nrows) continue;\n", "            if (r >= a.nrows || wave >= 8) continue;\n")
rep("        if (lane < cnt) {\n            const int r = stream_row(x, 8 * (c + lane) + wave, nx);\n            if (r < a.nrows)",
    "        if (lane < cnt && wave < 8) {\n            const int r = stream_row(x, 8 * (c + lane) + wave, nx);\n            if (r < a.nrows)")
rep("    const long nu_len = ctx->selfhalo ? (long)kExtMax : ctx->npix;", "    const long nu_len = ctx->selfhalo ? (long)kThreadPix : ctx->npix;")
if args.cf_lds:
    rep("""    double cF[VT_FDEG + 1];
#pragma unroll
    for (int k = 0; k <= VT_FDEG; ++k) cF[k] = tab[kZFLds + k];
""", "    const double* cF = tab + kZFLds;       // (read where they are used: 14 registers fewer across the line)\n")
for f in FILES:
    open(os.path.join(work, f), "w").write(text[f])
out = os.path.join(root, "build", "abl", "geom_%dx%d%s.so" % (args.threads, args.ppt, "_cflds" if args.cf_lds else ""))
log = bld.build_tree(work, out, stamp="geometry_experiment", report=True)
if args.report:
    for row in bld.resource_table(log):
        print("%-34s VGPR %3d scratch %3d occupancy %d sgpr-spill %3d vgpr-spill %3d" % ((row[0][-34:],) + row[1:]))
print("built", out)
