// What this chip's HBM gives a plain stream: fill (write only), read-sum (read only), copy (read + write), with hand-written
// kernels at several grid sizes and with the runtime's own hipMemsetAsync / hipMemcpyAsync.  The full-map kernel of the
// library is a WRITE stream (S^2 * 8 B per walker); this prices it.   build: hipcc -O3 --offload-arch=gfx950 -o hbm_rates hbm_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_fill(double2* __restrict__ dst, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = double2{v, v};
}
__global__ void __launch_bounds__(256) k_fill_nt(double2* __restrict__ dst, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        __builtin_nontemporal_store(v, &dst[i].x); __builtin_nontemporal_store(v, &dst[i].y);
    }
}
// every block fills its own contiguous share of the buffer
__global__ void __launch_bounds__(256) k_fill_chunk(double2* __restrict__ dst, size_t n, double v) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = (size_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) dst[i] = double2{v, v};
}
// ... four 16-byte stores per thread and trip (64 B per lane: a wave covers 4 KiB per trip)
__global__ void __launch_bounds__(256) k_fill_chunk4(double2* __restrict__ dst, size_t n, double v) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = (size_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + 256 * u < i1) dst[i + 256 * u] = double2{v, v};
    }
}
__global__ void __launch_bounds__(256) k_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_read(const double2* __restrict__ src, double* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const double2 v = src[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}

template <typename F> int timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch(); launch();
    CHK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-48s %8.3f ms  %7.0f GB/s\n", name, ms / reps, bytes * reps / ms * 1e-6);
    return 0;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;          // 2 GiB: the full-map kernel's bytes per launch at 512^2 x 1024 walkers
    double2 *a, *b; double* out;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&out, 8));
    CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
    char name[96];
    for (int bpc : {1, 2, 4, 8, 16, 32, 128}) {
        const int grid = 256 * bpc;
        snprintf(name, sizeof name, "fill, %d blocks per CU", bpc);
        timeit(name, (double)bytes, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n, 1.0); });
        snprintf(name, sizeof name, "fill (nontemporal), %d blocks per CU", bpc);
        timeit(name, (double)bytes, [&] { hipLaunchKernelGGL(k_fill_nt, dim3(grid), dim3(256), 0, 0, b, n, 1.0); });
        snprintf(name, sizeof name, "fill, contiguous share per block, %d blocks per CU", bpc);
        timeit(name, (double)bytes, [&] { hipLaunchKernelGGL(k_fill_chunk, dim3(grid), dim3(256), 0, 0, b, n, 1.0); });
        snprintf(name, sizeof name, "fill, contiguous share, 4 stores/trip, %d blocks per CU", bpc);
        timeit(name, (double)bytes, [&] { hipLaunchKernelGGL(k_fill_chunk4, dim3(grid), dim3(256), 0, 0, b, n, 1.0); });
        snprintf(name, sizeof name, "read, %d blocks per CU", bpc);
        timeit(name, (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n); });
        snprintf(name, sizeof name, "copy (read + write counted), %d blocks per CU", bpc);
        timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    }
    timeit("hipMemsetAsync", (double)bytes, [&] { (void)hipMemsetAsync(b, 0, bytes, 0); });
    timeit("hipMemcpyAsync device to device (read + write)", 2.0 * bytes, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
