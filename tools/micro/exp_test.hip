// exp_neg (voigt_device.h) against the device library's exp(-t), bit for bit.
// hipcc -O3 --offload-arch=gfx950 -std=c++17 -I mc-alf_amd/csrc -o /tmp/exp_test tools/micro/exp_test.hip && /tmp/exp_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "voigt_device.h"

__global__ void k(const double* t, double* a, double* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = mcalf::exp_neg(t[i]); b[i] = exp(-t[i]); }
}

int main() {
    std::vector<double> h;
    unsigned long long s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < 4000000; ++i) {
        const double u = rnd();
        const int kind = i & 7;
        double v;
        if (kind == 0) v = u * 1e-3;
        else if (kind == 1) v = u * 40.0;
        else if (kind == 2) v = u * 800.0;
        else if (kind == 3) v = -u * 720.0;
        else if (kind == 4) v = std::pow(10.0, -300.0 + 600.0 * u);
        else if (kind == 5) v = 700.0 + 400.0 * u;
        else if (kind == 6) v = -(700.0 + 400.0 * u);
        else v = u * 5.0;
        h.push_back(v);
    }
    const double special[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 745.13, 745.14, 1075.0, 1075.1, -709.78, -709.79, -1024.0, -1024.1, 708.3964, 1e-320, -1e-320};
    for (double v : special) h.push_back(v);
    const int n = (int)h.size();
    double *dt, *da, *db;
    hipMalloc(&dt, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dt, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dt, da, db, n);
    std::vector<double> a(n), b(n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        const bool same = (std::memcmp(&a[i], &b[i], 8) == 0) || (std::isnan(a[i]) && std::isnan(b[i]));
        if (!same && bad++ < 10) std::printf("t=%.17g  exp_neg=%.17g  exp=%.17g\n", h[i], a[i], b[i]);
    }
    std::printf("%d inputs, %ld differ\n", n, bad);
    return bad != 0;
}
