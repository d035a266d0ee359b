// Micro-benchmark: issue cost of v_mfma_f64_16x16x4_f64 on gfx950, by accumulator chains per wave and waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_rate mfma_f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
    v4d acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = v4d{seed, seed, seed, seed};
    const double a = seed * 1e-3 + threadIdx.x * 1e-6, b = seed * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u % CHAINS], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
int run(double* d, int blocks_per_cu) {
    const int iters = 2000, blocks = 256 * blocks_per_cu, threads = 256;      // one 4-wave block = 1 wave per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.5);
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.5);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)blocks * 4 * iters * 8 / (256.0 * 4);
    const double ns = ms * 1e6 / per_simd;
    printf("chains %d  waves/SIMD %d : %8.3f ms  %7.2f ns per MFMA per SIMD (%.0f cycles at 2.4 GHz)  -> %.1f TFLOP/s\n", CHAINS, blocks_per_cu,
           ms, ns, ns * 2.4, 2048.0 * per_simd * 1024 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    double* d; CHK(hipMalloc(&d, sizeof(double) * 256 * 8 * 256));
    for (int w = 1; w <= 4; w *= 2) { run<1>(d, w); run<2>(d, w); run<4>(d, w); }
    return 0;
}
