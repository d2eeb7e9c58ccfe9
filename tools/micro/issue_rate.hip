// Microbenchmark (diagnostic): cycles per wave-instruction of the instruction classes the fused kernel issues, on
// gfx950, at 1 wave per SIMD (the issue interval of ONE wave: nothing else competes, back-to-back independent
// instructions of the class) and at 4 waves per SIMD (what the pipe of that class sustains per SIMD with the kernel's
// occupancy).  Feeds bench.py's `roofline_issue` (profiles/pmc.json: issue.cycles).
//   hipcc -O3 --offload-arch=gfx950 -o build/issue_rate tools/micro/issue_rate.hip && build/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kPerIter = 64;     // instructions of the class per loop iteration

// MODE 0: v_fma_f64 (8 independent chains)      1: 32-bit VALU (v_add_u32, v_cndmask-like moves)
//      2: SALU (s_add_u32 on 8 registers)       3: LDS ds_read_b128 (broadcast address, 8 in flight)
//      4: s_cbranch_scc0 not taken + s_cmp (counted as 2 scalar instructions)
template <int MODE>
__global__ void k(double* out, unsigned long long* clk, int iters) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i;
    __syncthreads();
    double a[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-9 + i; u[i] = threadIdx.x + i; }
    const double b = 0.999999, c = 1e-3;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    const unsigned ldsaddr = (unsigned)(size_t)lds + (threadIdx.x & 7) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < kPerIter / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = fma(a[i], b, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < kPerIter / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < kPerIter / 8; ++r)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < kPerIter / 8; ++r) {
                v2d v0, v1, v2, v3;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                             "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(ldsaddr) : "memory");
                a[0] += v0.x + v1.x + v2.x + v3.x;                       // (4 extra VALU per 8 reads: subtracted below)
            }
        } else {
#pragma unroll
            for (int r = 0; r < kPerIter / 2; ++r)
                asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n 1:" : : "s"(s0) : "scc");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + u[i];
    s += s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    double* dout; unsigned long long* dclk;
    hipMalloc(&dout, 8 << 20); hipMalloc(&dclk, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* nm[5] = {"valu_f64", "valu_other", "salu", "lds", "branch"};
    printf("{\n");
    for (int wi = 0; wi < 2; ++wi) {
        const int wps = wi == 0 ? 1 : 4;               // waves per SIMD
        printf(" \"%s\": {", wps == 1 ? "wave" : "pipe");
        for (int m = 0; m < 5; ++m) {
            const int blocks = 256 * wps, threads = 256;     // 256 CUs x wps workgroups of 4 waves (one per SIMD)
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<blocks, threads>>>(dout, dclk, iters);
                if (m == 1) k<1><<<blocks, threads>>>(dout, dclk, iters);
                if (m == 2) k<2><<<blocks, threads>>>(dout, dclk, iters);
                if (m == 3) k<3><<<blocks, threads>>>(dout, dclk, iters);
                if (m == 4) k<4><<<blocks, threads>>>(dout, dclk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned long long h[2]; hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
            // clock held during the f64 run (s_memtime / s_memrealtime, the latter at 100 MHz); reused for the scalar
            // classes, whose s_memtime stamps are not ordered with the scalar ALU stream
            static double ghz = 2.4;
            if (m == 0) ghz = (double)h[0] / ((double)h[1] * 10.0);
            // SIMD time per wave-instruction from the kernel's duration (HIP events): with `wps` waves per SIMD the
            // SIMD issued wps * iters * kPerIter instructions of the class; at wps = 1 this is the issue interval of a wave
            const double per_simd_instr = (double)wps * iters * kPerIter;
            double per_instr = (double)ms * 1e-3 * ghz * 1e9 / per_simd_instr;
            if (m == 3) per_instr -= 0.5 * 4.4;                                      // half a VALU add rides on every read
            const double simd_view = per_instr;
            printf("%s\"%s\": %.3f", m ? ", " : "", nm[m], simd_view);
            if (m == 4) printf(", \"clock_ghz\": %.3f", ghz);
        }
        printf("}%s\n", wi == 0 ? "," : "");
    }
    printf("}\n");
    return 0;
}
