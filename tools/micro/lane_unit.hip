// What unit does SQ_THREAD_CYCLES_VALU count in?  Three kernels with a KNOWN share of active lanes -- all 64 lanes in a
// loop of v_fma_f64, 16 of 64 lanes in the same loop, all 64 lanes in a loop of v_fma_f32 -- to be run under
//   rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -- build/lane_unit
// tools/profiles_summary.py divides the fused kernel's thread-cycles by (vector instructions x 64 lanes x the unit measured
// here) to get the share of lanes that were enabled while the vector pipe worked (`valu_active_lane_fraction`).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void lanes64_f64(double* out, int n) {
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fma(a, b, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void lanes16_f64(double* out, int n) {
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
    if ((threadIdx.x & 63) < 16) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a = fma(a, b, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void lanes64_f32(float* out, int n) {
    float a = threadIdx.x * 1e-3f, b = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(a, b, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    const int blocks = 1024, threads = 256, n = 4096;
    double* d;
    float* f;
    if (hipMalloc(&d, sizeof(double) * blocks * threads) != hipSuccess || hipMalloc(&f, sizeof(float) * blocks * threads) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(lanes64_f64, dim3(blocks), dim3(threads), 0, 0, d, n);
        hipLaunchKernelGGL(lanes16_f64, dim3(blocks), dim3(threads), 0, 0, d, n);
        hipLaunchKernelGGL(lanes64_f32, dim3(blocks), dim3(threads), 0, 0, f, n);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::printf("{\"kernels\": [\"lanes64_f64\", \"lanes16_f64\", \"lanes64_f32\"], \"fma_per_lane\": %d}\n", 16 * n);
    return 0;
}
