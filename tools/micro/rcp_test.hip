// Microbenchmark (diagnostic): accuracy of v_rcp_f64 and relative issue cost of rcp vs fma.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k_rcp(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = fma(-v, r, 1.0); r = fma(r, e, r); r1[i] = r;
    e = fma(-v, r, 1.0); r = fma(r, e, r); r2[i] = r;
}
template <int MODE>
__global__ void k_time(double* out, double seed, int iters) {
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 0.5;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a = fma(a, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }
        if (MODE == 1) { a = __builtin_amdgcn_rcp(a) + 1.5; a2 = __builtin_amdgcn_rcp(a2) + 1.5; a3 = __builtin_amdgcn_rcp(a3) + 1.5; a4 = __builtin_amdgcn_rcp(a4) + 1.5; }
        if (MODE == 2) { a = a + b; a2 = a2 + b; a3 = a3 + b; a4 = a4 + b; }
        if (MODE == 3) { a = fmax(a, b) + c; a2 = fmax(a2, b)+c; a3 = fmax(a3, b)+c; a4 = fmax(a4, b)+c; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), r0(n), r1(n), r2(n);
    for (int i = 0; i < n; ++i) x[i] = 36.0 * pow(1e6 / 36.0, (double)i / n) * (1.0 + 1e-7 * (i % 977));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k_rcp<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / (long double)x[i];
        m0 = fmax(m0, fabs((double)((r0[i] - ex) / ex)));
        m1 = fmax(m1, fabs((double)((r1[i] - ex) / ex)));
        m2 = fmax(m2, fabs((double)((r2[i] - ex) / ex)));
    }
    printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n", m0, m1, m2);
    double* dout; hipMalloc(&dout, 256 * 1024 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000; const int blocks = 1024, threads = 256;   // 4 waves/SIMD
    const char* names[4] = {"fma_f64", "rcp_f64+add", "add_f64", "max+add f64"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) k_time<0><<<blocks, threads>>>(dout, 1.0, iters);
            if (m == 1) k_time<1><<<blocks, threads>>>(dout, 1.0, iters);
            if (m == 2) k_time<2><<<blocks, threads>>>(dout, 1.0, iters);
            if (m == 3) k_time<3><<<blocks, threads>>>(dout, 1.0, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr = (double)blocks * threads / 64 * iters * 4;
        // per SIMD: wave_instr / (256 CU * 4 SIMD)
        printf("%-14s %.3f ms  -> %.2f ns per wave-instr-group per SIMD\n", names[m], ms, ms * 1e6 / (wave_instr / 1024.0));
    }
    return 0;
}
