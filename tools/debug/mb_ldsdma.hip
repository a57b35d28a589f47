// hardware check of the direct global -> LDS loads of gfx950 (__builtin_amdgcn_global_load_lds): where does lane l's data land,
// do inactive lanes write, are 4-byte-aligned 16-byte sources accepted, does the instruction offset apply to the LDS side.
//   hipcc --offload-arch=gfx950 -O2 -o build/mb_ldsdma tools/debug/mb_ldsdma.hip ; gpurun -- ./build/mb_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void __launch_bounds__(64) k(const int* g, int* out, int mode) {
    __shared__ int lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1;
    __syncthreads();
    const int l = threadIdx.x;
    if (mode == 0) __builtin_amdgcn_global_load_lds(g + l * 100, lds + 256, 16, 0, 0);                 // 16 B per lane, aligned sources
    if (mode == 1) __builtin_amdgcn_global_load_lds(g + l * 100 + 1, lds + 256, 16, 0, 0);             // sources 4-byte aligned only
    if (mode == 2) { if (l & 1) __builtin_amdgcn_global_load_lds(g + l * 100, lds + 256, 16, 0, 0); }  // half of the lanes inactive
    if (mode == 3) __builtin_amdgcn_global_load_lds(g + l * 100, lds + 256, 4, 0, 0);                  // 4 B per lane
    if (mode == 4) __builtin_amdgcn_global_load_lds(g + l * 100, lds + 256, 12, 0, 0);                 // 12 B per lane
    if (mode == 5) __builtin_amdgcn_global_load_lds(g + l * 100, lds + 256, 16, 64, 0);                // instruction offset 64
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<int> h(64 * 100 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int)i;
    int *d_g, *d_out; CK(hipMalloc(&d_g, h.size() * 4)); CK(hipMalloc(&d_out, 2048 * 4));
    CK(hipMemcpy(d_g, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char* nm[6] = {"16 B, aligned", "16 B, source +4", "16 B, odd lanes only", "4 B", "12 B", "16 B, inst offset 64"};
    for (int mode = 0; mode < 6; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_g, d_out, mode);
        CK(hipDeviceSynchronize());
        std::vector<int> o(2048); CK(hipMemcpy(o.data(), d_out, 2048 * 4, hipMemcpyDeviceToHost));
        int first = -1, last = -1, n = 0;
        for (int i = 0; i < 2048; i++) if (o[i] != -1) { if (first < 0) first = i; last = i; n++; }
        printf("mode %d (%s): %d words written, LDS words [%d, %d]; first 12 from %d:", mode, nm[mode], n, first, last, first);
        for (int i = first; i < first + 12 && i >= 0 && i < 2048; i++) printf(" %d", o[i]);
        printf("\n");
    }
    return 0;
}
