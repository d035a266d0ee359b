// Resident blocks per CU of the stage-1 kernels at the launch shapes joxsz_hip.hip uses (hipOccupancyMaxActiveBlocksPerMultiprocessor).
//   hipcc -O2 --offload-arch=gfx950 -I joxsz_amd/csrc -o scripts/ubench/occ_rowmix scripts/ubench/occ_rowmix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "jx_mix.hpp"
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, LDS per CU %zu, per block %zu (optin %zu), regs per CU %d, max threads per CU %d\n", p.gcnArchName, p.multiProcessorCount,
           (size_t)p.maxSharedMemoryPerMultiProcessor, (size_t)p.sharedMemPerBlock, (size_t)p.sharedMemPerBlockOptin, p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor);
    auto k = jx_rowmix_mfma_kernel<8, double2>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k);
    printf("mfma kernel: %d VGPRs, %zu B static LDS, max threads %d\n", fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
    for (int threads : {256, 512, 1024}) for (size_t lds : {(size_t)40000, (size_t)58000, (size_t)74880, (size_t)107648}) {
        int n = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, threads, lds);
        printf("  threads %4d lds %6zu -> %d blocks/CU (%d waves/CU)\n", threads, lds, n, n * threads / 64);
    }
    auto k2 = jx_rowmix_kernel<18, 8, double2>;
    int n = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k2, 256, 18432);
    printf("valu kernel RT=18: %d blocks/CU of 256 threads\n", n);
    return 0;
}
