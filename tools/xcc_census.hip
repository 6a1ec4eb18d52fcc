// xcc_census.hip -- measurement aid: which XCD (HW_REG_XCC_ID) and CU does block i of a one-block-per-CU launch run on?
// Build: hipcc -O2 --offload-arch=gfx950 tools/xcc_census.hip -o /tmp/xcc_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void census(int *out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)xcc; out[2 * blockIdx.x + 1] = (int)hwid; }
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("multiProcessorCount %d arch %s\n", prop.multiProcessorCount, prop.gcnArchName);
    for (int threads : {64, 384}) {
        const int grid = prop.multiProcessorCount;
        int *d; hipMalloc(&d, 8 * grid);
        hipLaunchKernelGGL(census, dim3(grid), dim3(threads), 0, 0, d);
        std::vector<int> h(2 * grid);
        hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
        int bad = 0, cnt[16] = {0};
        for (int i = 0; i < grid; ++i) { cnt[h[2 * i] & 15]++; if ((h[2 * i] & 15) != (i & 7)) ++bad; }
        printf("threads %d grid %d: blocks not on XCD i%%8: %d; per-XCD counts:", threads, grid, bad);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n first 16 raw XCC_ID:");
        for (int i = 0; i < 16; ++i) printf(" %#x", h[2 * i]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
