// Do v_mfma_f64_16x16x4 and v_fma_f64 share the fp64 multipliers of a SIMD on gfx950, or do they run side by side?
// Three kernels, the same loop: (a) 4 independent matrix instructions per trip, (b) NF independent vector FMAs per trip, (c) both.
// If (c) takes max(a, b) the pipes are separate; if it takes a + b they are one.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o scripts/ubench/mfma_valu_overlap scripts/ubench/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE, int NF>
__global__ void __launch_bounds__(256) k(double* out, int trips, double a, double b) {
    v4d acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = v4d{0, 0, 0, 0};
    double f[NF];
    for (int i = 0; i < NF; ++i) f[i] = threadIdx.x * 1e-3 + i;
    double x = a + threadIdx.x, y = b;
    for (int it = 0; it < trips; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[t], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < NF; ++i) f[i] = fma(f[i], a, b);
        }
    }
    double s = 0;
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < NF; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NF> float run(double* d, int blocks, int trips) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NF><<<blocks, 256>>>(d, trips, 1.0000001, 1e-9);
    hipEventRecord(e0);
    k<MODE, NF><<<blocks, 256>>>(d, trips, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double* d; hipMalloc(&d, 8ull * 256 * 4096);
    const int trips = 4000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;                       // blocks of 4 waves: wps waves per SIMD
        const float a = run<1, 16>(d, blocks, trips), b16 = run<2, 16>(d, blocks, trips), c16 = run<3, 16>(d, blocks, trips);
        const float b4 = run<2, 4>(d, blocks, trips), c4 = run<3, 4>(d, blocks, trips);
        const double per = 1e6 / (double)trips / wps;       // ns per trip and wave slot
        printf("%d waves/SIMD: 4 MFMA %.3f ms (%.1f ns per trip per wave: %.1f clk per MFMA at 2.4 GHz) | 16 FMA %.3f ms (%.1f clk per FMA) | both %.3f ms (sum %.3f, max %.3f) || 4 FMA %.3f | 4 MFMA + 4 FMA %.3f (sum %.3f)\n",
               wps, a, a * per, a * per * 2.4 / 4, b16, b16 * per * 2.4 / 16, c16, a + b16, a > b16 ? a : b16, b4, c4, a + b4);
    }
    return 0;
}
