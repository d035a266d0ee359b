// Micro-benchmark: fp64 FMAs whose multiplier is a wave-uniform constant (SGPR operand), as in jx_rowmix_kernel.
//   mode 0: constants loaded once (no scalar loads in the loop)
//   mode 1: R constants re-loaded every step from a stationary address (scalar-cache hits)
//   mode 2: R constants streamed (address advances by R * 8 B per step; table of `rows` rows wraps)
// build: hipcc -O3 --offload-arch=gfx950 -o fma_sgpr fma_sgpr.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

//   NW: accumulator sets per lane that share each constant (jx_rowmix_kernel's walker sets per lane)
template <int R, int MODE, int NW>
__global__ void __launch_bounds__(256) k(const double* __restrict__ C, double* __restrict__ out, int steps, int rows, double seed) {
    double acc[NW][R];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[w][r] = seed + r + w;
    double f[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) f[w] = seed * 0.001 + threadIdx.x * 1e-6 + w;
    const double* cp = C + (size_t)(blockIdx.x % 7) * R;
    const double* cend = C + (size_t)rows * R;
    double c0[R];
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) c0[r] = cp[r];
    }
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int w = 0; w < NW; ++w) f[w] = fma(f[w], 0.999, 1e-9);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int w = 0; w < NW; ++w) acc[w][r] = fma(MODE == 0 ? c0[r] : cp[r], f[w], acc[w][r]);
        if (MODE == 2) { cp += R; if (cp >= cend) cp = C; }
    }
    double t = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int w = 0; w < NW; ++w) t += acc[w][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int R, int MODE, int NW = 1>
int run(const char* name, const double* C, double* out, int wps, int rows) {
    const int steps = 2000, blocks = 256 * wps, threads = 256;        // wps blocks per CU = wps waves per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<R, MODE, NW>), dim3(blocks), dim3(threads), 0, 0, C, out, 10, rows, 1.5);
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<R, MODE, NW>), dim3(blocks), dim3(threads), 0, 0, C, out, steps, rows, 1.5);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double fma = (double)blocks * threads * steps * (R + 1) * NW;
    printf("%-34s R=%2d NW=%d %d waves/SIMD: %7.3f ms  %6.2f T FMA/s  (%.0f%% of 39.3)\n", name, R, NW, wps, ms, fma / ms * 1e-9, fma / ms * 1e-9 / 39.3 * 100);
    return 0;
}

int main() {
    const int rows = 257;
    double *C, *out;
    CHK(hipMalloc(&C, sizeof(double) * rows * 64)); CHK(hipMalloc(&out, sizeof(double) * 256 * 8 * 256));
    std::vector<double> h(rows * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (i % 17);
    CHK(hipMemcpy(C, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    for (int wps : {1, 2, 4, 8}) {
        run<20, 0>("constants in registers", C, out, wps, rows);
        run<20, 1>("re-loaded each step, same address", C, out, wps, rows);
        run<20, 2>("streamed (160 B per step)", C, out, wps, rows);
        run<8, 2>("streamed (64 B per step)", C, out, wps, rows);
        run<32, 2>("streamed (256 B per step)", C, out, wps, rows);
        if (wps <= 4) {
            run<20, 0, 2>("constants in registers", C, out, wps, rows);
            run<20, 1, 2>("re-loaded each step, same address", C, out, wps, rows);
            run<20, 2, 2>("streamed (160 B per step)", C, out, wps, rows);
        }
        if (wps <= 2) run<20, 2, 3>("streamed (160 B per step)", C, out, wps, rows);
    }
    return 0;
}
