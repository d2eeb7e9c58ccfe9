// Micro-benchmark (GPU box): what a kernel pays for reading page-locked HOST memory over PCIe.
//   chase   one wave, N dependent 8-byte loads (latency of one round trip)
//   row     W waves, each reading ONE 47-double row per round with the strided pattern of setup_sample (3 loads per lane)
//   bulk    a workgroup copying a contiguous block host -> HBM, 6 coalesced loads in flight per thread
// hipcc -O3 --offload-arch=gfx950 -o /tmp/pcie_read tools/micro/pcie_read.hip && /tmp/pcie_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void chase(const unsigned long long* p, int n, unsigned long long* out) {
    unsigned long long i = 0;
    for (int k = 0; k < n; ++k) i = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) *out = i;
}

__global__ void rows(const double* p, int ndim, int rounds, double* out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = (gridDim.x * blockDim.x) >> 6;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const double* row = p + (size_t)(r * nw + wave) * ndim;
        const int q = 2 + 3 * (lane % 15);
        acc += row[q] + row[q + 1] + row[q + 2] + row[0] + row[1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 1.2345) out[0] = acc;
}

__global__ void bulk(const double* src, double* dst, int per_wg, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        const size_t base = ((size_t)r * gridDim.x + blockIdx.x) * per_wg;
        double v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { const int i = threadIdx.x + k * blockDim.x; v[k] = i < per_wg ? src[base + i] : 0.0; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { const int i = threadIdx.x + k * blockDim.x; if (i < per_wg) dst[base + i] = v[k]; }
        __syncthreads();
    }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const int ndim = 47, nrows = 4096;
    const size_t n = (size_t)nrows * ndim;
    double *h = nullptr, *d = nullptr, *dout = nullptr;
    for (int coherent = 1; coherent >= 0; --coherent) {
        CK(hipHostMalloc((void**)&h, n * sizeof(double), hipHostMallocMapped | (coherent ? hipHostMallocCoherent : hipHostMallocNonCoherent)));
        for (size_t i = 0; i < n; ++i) h[i] = 1.0 + i;
        CK(hipMalloc((void**)&d, n * sizeof(double)));
        CK(hipMalloc((void**)&dout, 64));
        unsigned long long* hc = (unsigned long long*)h;
        // pointer chain with a stride of 32 cache lines
        std::vector<unsigned long long> save(hc, hc + n);
        const int hops = 200; size_t idx = 0;
        for (int k = 0; k < hops; ++k) { size_t nxt = (idx + 32 * 8 + 8) % n; hc[idx] = nxt; idx = nxt; }
        float t = timeit([&] { hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, (const unsigned long long*)h, hops, (unsigned long long*)dout); }, 5);
        printf("%s host memory: dependent 8-byte load = %.2f us per hop (kernel of %d hops %.1f us)\n", coherent ? "coherent" : "non-coherent", t * 1e3 / hops, hops, t * 1e3);
        for (size_t i = 0; i < n; ++i) hc[i] = save[i];
        CK(hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, (const unsigned long long*)d, 0, (unsigned long long*)dout);
        for (int wgs : {1, 16, 64, 512}) {
            const int waves = wgs * 8, rounds = nrows / waves > 0 ? nrows / waves : 1;
            const int rr = rounds > 16 ? 16 : rounds;
            float th = timeit([&] { hipLaunchKernelGGL(rows, dim3(wgs), dim3(512), 0, 0, (const double*)h, ndim, rr, dout); }, 5);
            float td = timeit([&] { hipLaunchKernelGGL(rows, dim3(wgs), dim3(512), 0, 0, (const double*)d, ndim, rr, dout); }, 5);
            printf("  rows pattern, %3d workgroups x 8 waves, %2d rounds: host %.1f us (%.2f us per round), HBM %.1f us (%.2f per round)\n",
                   wgs, rr, th * 1e3, th * 1e3 / rr, td * 1e3, td * 1e3 / rr);
        }
        for (int wgs : {1, 16, 64}) {
            const int per_wg = 64 * ndim, rounds = nrows / 64 / wgs > 0 ? nrows / 64 / wgs : 1;
            float th = timeit([&] { hipLaunchKernelGGL(bulk, dim3(wgs), dim3(512), 0, 0, (const double*)h, d, per_wg, rounds); }, 5);
            printf("  bulk copy host->HBM, %2d workgroups x %d rounds of 64 rows (24 KB): %.1f us = %.2f us per round, %.1f GB/s\n",
                   wgs, rounds, th * 1e3, th * 1e3 / rounds, (double)wgs * rounds * per_wg * 8 / (th * 1e-3) / 1e9);
        }
        hipFree(d); hipFree(dout); hipHostFree(h);
    }
    return 0;
}
