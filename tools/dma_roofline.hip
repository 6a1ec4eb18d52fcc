// dma_roofline.hip -- measurement aid (not part of the library): the sustained global->LDS rate of one CU
// for the access pattern of the conv producers (16-byte lanes, 1 KB per wave instruction, L2-resident
// source), as a function of waves per CU and LDS-DMA instructions in flight per wave, for the flat
// (global_load_lds) and the buffer (buffer_load ... lds) forms, with and without concurrent MFMA waves.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma_roofline.hip -o /tmp/dma_roofline ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LOADERS waves stream `iters` x INFLIGHT KB each through a private 16 KB LDS window; MFMAW further waves
// issue back-to-back MFMAs (independent accumulators) for the whole time.
template <int INFLIGHT, bool BUF>
__global__ __launch_bounds__(1024) void dma_kernel(const char *src, unsigned src_bytes, int loaders, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave < loaders) {
        unsigned char *base = smem + wave * (INFLIGHT * 1024);
        // every wave of every workgroup walks the same 2 MB (L2-resident) in 1 KB steps, offset by its id
        unsigned off = (unsigned)(((blockIdx.x * 16 + wave) * 4099u) % 2048u) * 1024u + lane * 16;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, src_bytes, 0x00020000);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                if (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(base + j * 1024), 16, (int)off, 0, 0, 0);
                else __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + off), (lds_void_t *)(base + j * 1024), 16, 0, 0);
                off = (off + 64 * 1024) & (2048u * 1024u - 1);
            }
            wait_vmcnt<INFLIGHT / 2>();          // keep half of the window in flight
        }
        wait_vmcnt<0>();
    } else {
        floatx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = (float)(lane + i);
        half8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        for (int it = 0; it < iters * INFLIGHT / 2; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 123.456f) sink[0] = s;
    }
}

template <int INFLIGHT, bool BUF>
static double run(const char *src, unsigned bytes, int wgs_per_cu, int loaders, int mfmaw, int iters, float *sink) {
    const int threads = 64 * (loaders + mfmaw);
    const size_t lds = (size_t)loaders * INFLIGHT * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((dma_kernel<INFLIGHT, BUF>), dim3(grid), dim3(threads), lds, 0, src, bytes, loaders, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_kernel<INFLIGHT, BUF>), dim3(grid), dim3(threads), lds, 0, src, bytes, loaders, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)grid * loaders * iters * INFLIGHT * 1024.0;
    return total / (ms * 1e-3) / 1e12;          // TB/s chip-wide
}

int main() {
    const unsigned bytes = 4u << 20;
    char *src; float *sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 64);
    printf("form   wg/CU loaders mfma_waves inflight/wave   TB/s   B/clk/CU@2.1GHz\n");
    const int iters = 400;
    struct C { int wg, ld, mf; };
    const C cfgs[] = {{1, 4, 0}, {2, 4, 0}, {1, 8, 0}, {2, 8, 0}, {1, 4, 4}, {2, 4, 4}, {1, 8, 8}};
    for (auto c : cfgs) {
        double r;
#define ROW(N, B)                                                                                     \
        r = run<N, B>(src, bytes, c.wg, c.ld, c.mf, iters, sink);                                       \
        printf("%-6s %5d %7d %10d %13d %7.2f %10.1f\n", B ? "buffer" : "global", c.wg, c.ld, c.mf, N, r, \
               r * 1e12 / 256 / 2.1e9);
        ROW(4, false) ROW(8, false) ROW(16, false) ROW(4, true) ROW(8, true) ROW(16, true)
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
