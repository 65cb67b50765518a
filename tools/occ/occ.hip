// Resident blocks per CU as a function of dynamic LDS bytes (320-thread blocks): finds the LDS allocation granule.
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ __launch_bounds__(320) void k(float* p) {
    extern __shared__ float s[];
    s[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    p[threadIdx.x] = s[(threadIdx.x + 1) % 320];
}
int main() {
    for (int lds : {54613, 65536, 73728, 75000, 77760, 78848, 79872, 80352, 80896, 81920}) {
        int n = 0;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 320, lds);
        printf("lds %6d -> %d blocks/CU (%s)\n", lds, n, hipGetErrorString(e));
    }
    return 0;
}
