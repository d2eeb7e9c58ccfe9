// Microbenchmark (diagnostic): sustained issue rate of f64/f32 FMA on gfx950 and the clock held.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, unsigned long long* clk, int iters) {
    double a[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-9 + i; f[i] = (float)a[i]; }
    const double b = 0.999999, c = 1e-3; const float bf = 0.999999f, cf = 1e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = fma(a[i], b, c);
            if (MODE == 1) f[i] = fmaf(f[i], bf, cf);
            if (MODE == 2) a[i] = a[i] * b;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    double* dout; unsigned long long* dclk; hipMalloc(&dout, 8 << 20); hipMalloc(&dclk, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    const char* nm[3] = {"fma_f64", "fma_f32", "mul_f64"};
    for (int wps = 1; wps <= 8; wps *= 2) {           // waves per SIMD
        for (int m = 0; m < 3; ++m) {
            int blocks = 256 * wps, threads = 256;     // 256 CUs x wps blocks of 4 waves
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<blocks, threads>>>(dout, dclk, iters);
                if (m == 1) k<1><<<blocks, threads>>>(dout, dclk, iters);
                if (m == 2) k<2><<<blocks, threads>>>(dout, dclk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned long long h[2]; hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
            double winstr_per_simd = (double)wps * iters * 8;
            double ghz = (double)h[0] / ((double)h[1] * 10.0);   // memrealtime = 100 MHz
            printf("%s waves/SIMD=%d: %.3f ms, %.2f ns per wave-instr per SIMD, clock %.2f GHz -> %.2f cycles/instr, %.1f TFLOP/s\n",
                   nm[m], wps, ms, ms * 1e6 / winstr_per_simd, ghz, ms * 1e6 / winstr_per_simd * ghz,
                   2.0 * 64 * winstr_per_simd * 1024 / (ms * 1e-3) / 1e12 * (m == 2 ? 0.5 : 1.0));
        }
    }
    return 0;
}
