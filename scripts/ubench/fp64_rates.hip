// Micro-benchmark: per-wave issue cost of fp64 vector instructions on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 -o fp64_rates fp64_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
    double a[8];
    for (int u = 0; u < 8; ++u) a[u] = seed + threadIdx.x * 1e-3 + u;
    const double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) a[u] = fma(a[u], b, c);
            if (OP == 1) a[u] = a[u] * b;
            if (OP == 2) a[u] = a[u] + c;
            if (OP == 3) a[u] = __builtin_amdgcn_rsq(a[u]) + c;
            if (OP == 4) a[u] = __builtin_amdgcn_rcp(a[u]) + c;
            if (OP == 5) a[u] = __builtin_amdgcn_sqrt(a[u]) + c;
            if (OP == 6) a[u] = 1.0 / sqrt(a[u]) + c;
            if (OP == 7) a[u] = pow(a[u], b) * 1e-3 + c;
        }
    }
    double s = 0;
    for (int u = 0; u < 8; ++u) s += a[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
int run(const char* name, double* d, int extra) {
    const int iters = 2000, blocks = 256 * 8, threads = 256;      // 8 blocks/CU = 8 waves/SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.5);
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.5);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = blocks*4 waves * iters*8 / (256 CUs * 4 SIMDs)
    const double winst = (double)blocks * 4 * iters * 8 / (256.0 * 4);
    const double ns_per = ms * 1e6 / winst;
    printf("%-22s %8.3f ms  %7.2f ns per wave-op per SIMD (= %.1f cycles at 2.4 GHz, incl. %d helper op)\n", name, ms, ns_per, ns_per * 2.4, extra);
    return 0;
}

int main() {
    double* d; CHK(hipMalloc(&d, sizeof(double) * 256 * 8 * 256));
    run<0>("v_fma_f64", d, 0); run<1>("v_mul_f64", d, 0); run<2>("v_add_f64", d, 0);
    run<3>("v_rsq_f64 (+add)", d, 1); run<4>("v_rcp_f64 (+add)", d, 1); run<5>("v_sqrt_f64 (+add)", d, 1);
    run<6>("1.0/sqrt(x) (+add)", d, 1); run<7>("pow(x,y) (+fma)", d, 1);
    return 0;
}
