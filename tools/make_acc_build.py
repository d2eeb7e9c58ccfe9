#!/usr/bin/env python3
"""Diagnostic: build `build/abl/stamps.so`, a copy of the library whose fused kernel accumulates, per wave, the
`s_memtime` cycles spent in the fold, at the loop barriers, in the node pass and in the direct evaluations
(the numbers quoted in DESIGN.md section 8).  The product source is patched in a scratch directory, not edited.

    python tools/make_acc_build.py && MCALF_HIP_LIB=build/abl/stamps.so python tools/stamp_report.py B 1024
"""
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mc-alf_amd", "csrc")
work = os.path.join(root, "build", "acc_src")
os.makedirs(os.path.join(work, "include"), exist_ok=True)
os.makedirs(os.path.join(root, "build", "abl"), exist_ok=True)
for f in ("voigt_device.h", "voigt_tables.h"):
    shutil.copy(os.path.join(src, f), work)
shutil.copy(os.path.join(root, "include", "mcalf_hip.h"), os.path.join(work, "include"))
s = open(os.path.join(src, "mcalf_hip.hip")).read().replace('"../../include/mcalf_hip.h"', '"include/mcalf_hip.h"')


def rep(a, b):
    global s
    if s.count(a) < 1:
        sys.exit("make_acc_build: the source no longer contains:\n" + a)
    s = s.replace(a, b, 1)


rep("template <bool kZeroPad, bool kSelfHalo, int kLinesPerSync, bool kInline>\n__global__ __launch_bounds__(kBlock, MCALF_MIN_WAVES) void mcalf_fused_kernel(const KArgs a) {",
    "__device__ unsigned long long g_acc[8192 * 32];\n#define CLK() __builtin_amdgcn_s_memtime()\n"
    "template <bool kZeroPad, bool kSelfHalo, int kLinesPerSync, bool kInline>\n__global__ __launch_bounds__(kBlock, MCALF_MIN_WAVES) void mcalf_fused_kernel(const KArgs a) {\n"
    "    unsigned long long accFold = 0, accBar = 0, accNode = 0, accDirect = 0;")
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
rep("        MCALF_STAMP(3);",
    "        MCALF_STAMP(3);\n        if ((threadIdx.x & 63) == 0 && w < 8192) { unsigned long long* q = g_acc + w * 32 + 4 * (threadIdx.x >> 6); "
    "q[0] = accFold; q[1] = accBar; q[2] = accNode; q[3] = accDirect; }\n        accFold = accBar = accNode = accDirect = 0;")
rep('extern "C" int mcalf_diag_read_dbg',
    'extern "C" int mcalf_diag_read_acc(unsigned long long* out, int n) {\n'
    '    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mcalf::g_acc), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;\n}\n'
    'extern "C" int mcalf_diag_read_dbg')
open(os.path.join(work, "acc.hip"), "w").write(s)
out = os.path.join(root, "build", "abl", "stamps.so")
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DMCALF_STAMPS=1"] + (["-DMCALF_COUNT_INTERP=1"] if os.environ.get("MCALF_COUNT_INTERP") else []) + ["-o", out, "acc.hip"]
subprocess.check_call(cmd, cwd=work)
print("built", out)
