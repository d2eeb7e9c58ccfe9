#!/usr/bin/env python3
"""Diagnostic builds of the library.  The product source carries NO instrumentation and no ablation switches: this script
patches a copy of it in a scratch directory (build/acc_src) and compiles that into build/abl/<name>.so, to be selected
with MCALF_HIP_LIB.

    python tools/make_acc_build.py                       # build/abl/stamps.so: phase timestamps per work item + per-wave
                                                         # s_memtime accumulators (fold / barrier / node pass / direct)
    python tools/make_acc_build.py --count-interp        # ... plus counters of interpolated / seen segments
    python tools/make_acc_build.py --name noloop --no-stamps --abl-noloop          # component loop removed
    python tools/make_acc_build.py --name setup_abl1 --no-stamps --abl-setup 1     # set-up without taps (2: without records, 3: neither)
    python tools/make_acc_build.py --name nointerp --no-stamps --no-far-interp     # every pixel evaluated directly

    MCALF_HIP_LIB=build/abl/stamps.so python tools/stamp_report.py B 1024
    MCALF_HIP_LIB=build/abl/stamps.so python tools/timeline_report.py C 4096

Every patch names the exact source text it hooks on and fails loudly when that text has changed."""
import argparse
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mc-alf_amd", "csrc")
work = os.path.join(root, "build", "acc_src")

ap = argparse.ArgumentParser()
ap.add_argument("--name", default="stamps")
ap.add_argument("--no-stamps", action="store_true")
ap.add_argument("--count-interp", action="store_true")
ap.add_argument("--abl-noloop", action="store_true")
ap.add_argument("--abl-setup", type=int, default=0, help="bit 0: no taps, bit 1: no records")
ap.add_argument("--no-far-interp", action="store_true")
args = ap.parse_args()
if os.environ.get("MCALF_COUNT_INTERP"):
    args.count_interp = True

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
        sys.exit("make_acc_build: expected %d occurrence(s), found %d, of:\n%s" % (count, found, a))
    for f in FILES:
        text[f] = text[f].replace(a, b)


STAMP = ("do { if (threadIdx.x == 0 && w < 8192) g_stamps[w * 8 + (%d)] = %s; } while (0);")


def stamp(k):
    return STAMP % (k, "__builtin_amdgcn_s_memrealtime()" if k in (0, 7) else "__builtin_amdgcn_s_memtime()")


if not args.no_stamps or args.count_interp:
    rep("// acc += a * b and acc += a with the accumulator tied to its register",
        "__device__ unsigned long long g_stamps[8192 * 8];\n__device__ unsigned long long g_dbg[4];   // [0] interpolated segments, "
        "[1] segments seen, [2] interpolable\n__device__ unsigned long long g_acc[8192 * 32];\n#define CLK() __builtin_amdgcn_s_memtime()\n"
        "// acc += a * b and acc += a with the accumulator tied to its register")
    rep('// ---- kernel entry points for the host files (kernel_args.h) -------------------------------------------------------------',
        'extern "C" int mcalf_diag_read_stamps(unsigned long long* out, int n) {\n'
        '    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcalf::g_stamps), (size_t)n * sizeof(unsigned long long));\n}\n'
        'extern "C" int mcalf_diag_read_dbg(unsigned long long* out) {\n'
        '    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcalf::g_dbg), 4 * sizeof(unsigned long long));\n}\n'
        'extern "C" int mcalf_diag_read_acc(unsigned long long* out, int n) {\n'
        '    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mcalf::g_acc), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;\n}\n'
        '// ---- kernel entry points for the host files (kernel_args.h) -------------------------------------------------------------')

if args.count_interp:
    rep("        done = uniform64((mp | mn) & segOk);                 // (segOk carries bits 8j only, so `done` does too)\n",
        "        done = uniform64((mp | mn) & segOk);\n"
        "        if ((threadIdx.x & 63) == 0) { atomicAdd(&g_dbg[0], (unsigned long long)__popcll(done)); atomicAdd(&g_dbg[1], 8ULL); "
        "atomicAdd(&g_dbg[2], (unsigned long long)__popcll(segOk)); }\n")

if not args.no_stamps:
    # phase timestamps of every work item (thread 0; `w` = the item index in scope): 0 item begins, 1 set-up data in LDS,
    # 2 first barrier passed, 3 component loop done, 4 flux tile published, 5 terms done, 6 slot, 7 item ends
    rep("    ItemLoads L;\n    request_item<kZeroPad, kSelfHalo, kInline, kStream>(a, w, tid0, L);\n\n    while (true) {\n",
        "    ItemLoads L;\n    request_item<kZeroPad, kSelfHalo, kInline, kStream>(a, w, tid0, L);\n"
        "    unsigned long long accFold = 0, accBar = 0, accNode = 0, accDirect = 0;\n\n    while (true) {\n        " + stamp(0) + "\n")
    rep("        // ---- 2. tau for this thread's pixels ----", "        " + stamp(1) + "\n        // ---- 2. tau for this thread's pixels ----")
    rep("        int buf = 0;\n        const int ncl_run = ncl;\n", "        " + stamp(2) + "\n        int buf = 0;\n        const int ncl_run = ncl;\n")
    rep("        if (hd.ngeneral > 0 && ncl_run > 0) eval_general_lines(sRec, ncl, nu, tau);\n",
        "        if (hd.ngeneral > 0 && ncl_run > 0) eval_general_lines(sRec, ncl, nu, tau);\n        " + stamp(3) + "\n"
        "        if ((threadIdx.x & 63) == 0 && w < 8192) { unsigned long long* q = g_acc + w * 32 + 4 * (threadIdx.x >> 6); "
        "q[0] = accFold; q[1] = accBar; q[2] = accNode; q[3] = accDirect; }\n        accFold = accBar = accNode = accDirect = 0;\n")
    rep("        // ---- 3+4. convolution, continuum, likelihood terms ----", "        " + stamp(4) + "\n        // ---- 3+4. convolution, continuum, likelihood terms ----")
    rep("        const bool more = tNext < nItems;\n", "        const bool more = tNext < nItems;\n        " + stamp(5) + "\n")
    rep("            if (a.ntiles == 1) {\n",
        "            do { if (threadIdx.x == 0 && w < 8192) g_stamps[w * 8 + 6] = blockIdx.x; } while (0);\n            " + stamp(7) + "\n"
        "            if (a.ntiles == 1) {\n")
    # per-wave cycle accumulators inside the component loop
    rep("                                          double& farNode, unsigned long long segOk) {\n    const double A = tab[kLineLds + 1], B = tab[kLineLds + 2];",
        "                                          double& farNode, unsigned long long segOk, unsigned long long& accNode, unsigned long long& accDirect) {\n"
        "    const unsigned long long c0 = __builtin_amdgcn_s_memtime();\n    const double A = tab[kLineLds + 1], B = tab[kLineLds + 2];")
    rep("    const unsigned doneLo = (unsigned)done, doneHi = (unsigned)(done >> 32);",
        "    const unsigned long long c1 = __builtin_amdgcn_s_memtime();\n    accNode += c1 - c0;\n"
        "    const unsigned doneLo = (unsigned)done, doneHi = (unsigned)(done >> 32);")
    rep("        fmac_inplace(tau[j], t, P);\n    }\n    }\n}", "        fmac_inplace(tau[j], t, P);\n    }\n    }\n    accDirect += __builtin_amdgcn_s_memtime() - c1;\n}")
    rep("eval_line(tabs + l * kTabPad, nu, tau, nuNode, farNode, segOk);",
        "eval_line(tabs + l * kTabPad, nu, tau, nuNode, farNode, segOk, accNode, accDirect);")
    rep("            double* tabs = sTab + buf * (kLinesPerSync * kTabPad);\n            if (hasCoef) {",
        "            const unsigned long long f0 = CLK();\n            double* tabs = sTab + buf * (kLinesPerSync * kTabPad);\n            if (hasCoef) {")
    rep("            __syncthreads();\n            buf ^= 1;",
        "            const unsigned long long f1 = CLK();\n            __syncthreads();\n            const unsigned long long f2 = CLK();\n"
        "            accFold += f1 - f0; accBar += f2 - f1;\n            buf ^= 1;")

if args.abl_noloop:
    rep("        const int ncl_run = ncl;\n", "        const int ncl_run = 0;       // ABLATION: no component loop\n")
if args.abl_setup & 2:
    rep("    for (int slot = lane; slot < nSlots; slot += 64) {\n", "    for (int slot = lane; slot < 0; slot += 64) {      // ABLATION: no records\n")
if args.abl_setup & 1:
    rep("    if (ntap8 <= 64) {                               // the usual case: one tap per lane, one exp\n",
        "    if (false) {                                     // ABLATION: no taps\n")
    rep("    } else {\n        double gsum = 0.0;\n", "    } else if (false) {\n        double gsum = 0.0;\n")
    rep("    const double bot = kZeroPad ? 1.0 : botOrdered;\n    const int ngen = ngenLane;\n", "    const double bot = 1.0;\n    const int ngen = 0;\n")
if args.no_far_interp:
    rep("constexpr bool kFarInterp = true;", "constexpr bool kFarInterp = false;")

for f in FILES:
    open(os.path.join(work, f), "w").write(text[f])
out = os.path.join(root, "build", "abl", args.name + ".so")
bld.build_tree(work, out, stamp="instrumented:" + args.name)
print("built", out)
