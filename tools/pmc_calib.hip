// pmc_calib.hip -- known-size traffic for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns this
// library uses (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of a wide coalesced streaming read ... other
// access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//   store16 / store8 : every lane stores 16 / 8 bytes, consecutive lanes consecutive addresses (the NHWC epilogues)
//   load16           : every lane loads 16 bytes (buffer_load_dwordx4-like), result folded into one store per block
//   store_nt16       : non-temporal 16-byte stores (the NCHW f32 mask logits)
// Each kernel moves exactly BYTES bytes (256 MiB, beyond the 256 MB Infinity Cache only marginally -- the counters sit on the
// L2's fabric side, so cache hits downstream of them are included either way).  Build + run: tools/measure/gpu_pmc_calib.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr size_t BYTES = 256ull << 20;
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

__global__ void store16(uint4v *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4v v = {(unsigned)i, 1u, 2u, 3u};
        p[i] = v;
    }
}
__global__ void store8(uint2v *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint2v v = {(unsigned)i, 1u};
        p[i] = v;
    }
}
__global__ void store_nt16(uint4v *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4v v = {(unsigned)i, 1u, 2u, 3u};
        __builtin_nontemporal_store(v, p + i);
    }
}
__global__ void load16(const uint4v *p, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4v v = p[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;      // (keeps the loads alive; practically never true)
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    void *a = nullptr, *b = nullptr;
    unsigned *sink = nullptr;
    CK(hipMalloc(&a, BYTES));
    CK(hipMalloc(&b, BYTES));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(a, 1, BYTES));
    CK(hipMemset(b, 1, BYTES));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 8, block = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(store16, dim3(grid), dim3(block), 0, 0, (uint4v *)a, BYTES / 16);
        hipLaunchKernelGGL(store8, dim3(grid), dim3(block), 0, 0, (uint2v *)b, BYTES / 8);
        hipLaunchKernelGGL(store_nt16, dim3(grid), dim3(block), 0, 0, (uint4v *)a, BYTES / 16);
        hipLaunchKernelGGL(load16, dim3(grid), dim3(block), 0, 0, (const uint4v *)b, BYTES / 16, sink);
        CK(hipDeviceSynchronize());
    }
    printf("pmc_calib: every kernel moved %zu bytes per launch, 3 launches each\n", BYTES);
    return 0;
}
