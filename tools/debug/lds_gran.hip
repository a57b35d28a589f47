// how many 64-thread workgroups with a given LDS size fit on one CU (LDS allocation granularity / per-CU budget)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(int* o) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[blockIdx.x] = s[(threadIdx.x + 1) & 63]; }
int main() {
    for (int b : {4096, 5120, 8192, 8448, 8560, 8704, 8705, 9216, 9217, 9584, 10240, 10241, 21760, 23552, 24576}) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, (size_t)b);
        printf("dyn LDS %6d B -> %2d workgroups per CU (%s) -> %d B total\n", b, n, hipGetErrorString(e), n * b);
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerBlock %zu regsPerMultiprocessor %d CUs %d clock %d kHz\n", p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerMultiprocessor, p.multiProcessorCount, p.clockRate);
    return 0;
}
