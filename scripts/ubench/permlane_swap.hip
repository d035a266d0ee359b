// What v_permlane32_swap / v_permlane16_swap (gfx950) do to two registers, lane by lane, and the 4 x 4 transpose of 16-lane
// rows that stage 1 on the matrix cores builds from them (jx_mix.hpp).
//   hipcc -O2 --offload-arch=gfx950 -o scripts/ubench/permlane_swap scripts/ubench/permlane_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r.x; out[64 + l] = r.y;
    r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = r.x; out[192 + l] = r.y;
    // transpose: f[k] = 1000 k + lane  ->  want b[t] lane (16 k + i) = f[k] lane (16 t + i) = 1000 k + 16 t + i
    unsigned f0 = l, f1 = 1000 + l, f2 = 2000 + l, f3 = 3000 + l;
    r = __builtin_amdgcn_permlane32_swap(f0, f2, false, false); f0 = r.x; f2 = r.y;
    r = __builtin_amdgcn_permlane32_swap(f1, f3, false, false); f1 = r.x; f3 = r.y;
    r = __builtin_amdgcn_permlane16_swap(f0, f1, false, false); f0 = r.x; f1 = r.y;
    r = __builtin_amdgcn_permlane16_swap(f2, f3, false, false); f2 = r.x; f3 = r.y;
    out[256 + l] = f0; out[320 + l] = f1; out[384 + l] = f2; out[448 + l] = f3;
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 4);
    k<<<1, 64>>>(d);
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[8] = {"swap32 .x", "swap32 .y", "swap16 .x", "swap16 .y", "b0", "b1", "b2", "b3"};
    for (int r = 0; r < 8; ++r) { printf("%-10s", nm[r]); for (int g = 0; g < 4; ++g) printf(" [%4u..%4u]", h[64 * r + 16 * g], h[64 * r + 16 * g + 15]); printf("\n"); }
    int ok = 1;
    for (int t = 0; t < 4; ++t) for (int l = 0; l < 64; ++l) if (h[256 + 64 * t + l] != 1000u * (l >> 4) + 16u * t + (l & 15)) ok = 0;
    printf("transpose %s\n", ok ? "ok" : "WRONG");
    return 0;
}
