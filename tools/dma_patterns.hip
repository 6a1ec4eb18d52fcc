// dma_patterns.hip -- measurement aid (not part of the library).  tools/dma_roofline.hip measured the global->LDS rate of a
// CU for CONTIGUOUS 1 KB wave instructions (42 B/clk/CU beside MFMA waves); the convolution producers reach about half of
// that.  Their access pattern differs in three ways, and this probe prices each one separately:
//   rows    -- one wave instruction = 8 activation rows x 128 B (8 different cache lines, row stride = channels x 2 B)
//              instead of 1 KB contiguous;
//   swizzle -- inside a row the eight 16-byte lanes read the line in XOR-permuted order (the LDS-DMA writes lane-linear,
//              so the bank-conflict swizzle of the fragment reads has to be applied on the SOURCE address):
//                 0 none, 1 slot ^= (row >> 1) & 7 (what conv_igemm / conv_wreg do), 2 slot ^= 4 * ((row >> 1) & 1)
//                 (whole 64-byte halves swap: every lane quad stays ascending), 3 slot = (slot + 2 * row) & 7 (rotation);
//   shared  -- the 32 workgroups of an XCD read the SAME bytes at the same time (a weight panel / an activation tile with
//              two N tiles) instead of private streams.
// Also: the same gather into VGPRs (buffer_load_dwordx4) instead of LDS, and sc1 (L2-served) loads.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma_patterns.hip -o /tmp/dma_patterns ; run on an MI355X (a few seconds).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Cfg {
    int rows;        // 0: 1 KB contiguous per instruction; 1: 8 rows x 128 B
    int stride;      // row stride in bytes (rows = 1)
    int swz;         // 0..3, see above
    int shared;      // 1: every workgroup walks the same addresses
    int vgpr;        // 1: buffer_load_dwordx4 into registers instead of LDS-DMA; 2: ... and from there to LDS with ds_write_b128
    int aux;         // cache policy bits of the load (0 default, 16 sc1)
    int loaders, mfmaw, iters;
};

// every loader wave streams iters x INFLIGHT instructions; the region is 2 MB (L2-resident after the first touch)
template <int INFLIGHT, int VGPR, int AUX>
__global__ __launch_bounds__(1024) void pat_kernel(const char *src, unsigned src_bytes, Cfg c, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    volatile int *done = (volatile int *)(smem + c.loaders * (INFLIGHT * 1024));    // loader waves finished (the MFMA waves run until then)
    if (threadIdx.x == 0) *done = 0;
    __syncthreads();
    if (wave < c.loaders) {
        unsigned char *base = smem + wave * (INFLIGHT * 1024);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, src_bytes, 0x00020000);
        const unsigned region = 2048u * 1024u;
        const unsigned who = c.shared ? (unsigned)wave : (unsigned)(blockIdx.x * 16 + wave);
        uint4v acc = {0, 0, 0, 0};
        uint4v stg[2][INFLIGHT / 2];
        // address generators: contiguous 1 KB steps, or (tile of 64 rows) x (K tiles of 128 B along the row)
        unsigned off_lin = ((who * 4099u) % 2048u) * 1024u + lane * 16;
        const int r8 = lane >> 3, slot = lane & 7;
        const unsigned m1 = c.swz == 1 ? ~0u : 0u, m2 = c.swz == 2 ? ~0u : 0u, m3 = c.swz == 3 ? ~0u : 0u;
        const unsigned stride = c.rows ? (unsigned)c.stride : 128u;
        const unsigned nrows = region / stride, ktiles = stride / 128u;
        unsigned tile = (who * 37u) % (nrows / 64u), kt = 0, j8 = 0;
        auto next_off = [&]() -> unsigned {
            if (!c.rows) {
                const unsigned o = off_lin;
                off_lin = (off_lin + 64 * 1024) & (region - 1);
                return o;
            }
            const unsigned row = tile * 64u + j8 * 8u + (unsigned)r8;
            // branch-free selection of the permutation (m1 / m2 / m3 are all-ones for the chosen one)
            const int s = ((slot ^ (int)(((row >> 1) & 7u) & m1) ^ (int)((((row >> 1) & 1u) * 4u) & m2)) +
                           (int)((2u * (row & 3u)) & m3)) & 7;
            const unsigned o = row * stride + kt * 128u + (unsigned)s * 16u;
            if (++j8 == 8u) {                               // 8 instructions = one 64-row K tile; then the next K tile
                j8 = 0;
                if (++kt == ktiles) { kt = 0; tile = (tile + 1u) % (nrows / 64u); }
            }
            return o;
        };
        for (int it = 0; it < c.iters; ++it) {
            if (VGPR == 2) {
                // the a_stage data path: half of the window is in flight while the other half is written to LDS
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int j = 0; j < INFLIGHT / 2; ++j)
                        stg[h][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)next_off(), 0, AUX);
#pragma unroll
                    for (int j = 0; j < INFLIGHT / 2; ++j)
                        *(uint4v *)(base + (h * (INFLIGHT / 2) + j) * 1024 + ((lane ^ (j & 7)) * 16)) = stg[h ^ 1][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) {
                    const unsigned off = next_off();
                    if (VGPR == 1) acc += __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, AUX);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(base + j * 1024), 16, (int)off, 0, 0, AUX);
                }
                if (!VGPR) wait_vmcnt<INFLIGHT / 2>();
            }
        }
        wait_vmcnt<0>();
        if (VGPR == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[1] = 1.f;
        if (lane == 0) atomicAdd((int *)done, 1);
    } else {
        floatx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = (float)(lane + i);
        half8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        while (*done < c.loaders) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 123.456f) sink[0] = s;
    }
}

static double run(const char *src, unsigned bytes, const Cfg &c, float *sink) {
    constexpr int INFLIGHT = 8;
    const int threads = 64 * (c.loaders + c.mfmaw);
    const size_t lds = (size_t)c.loaders * INFLIGHT * 1024 + 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (c.vgpr == 2) hipLaunchKernelGGL((pat_kernel<INFLIGHT, 2, 0>), dim3(256), dim3(threads), lds, 0, src, bytes, c, sink);
        else if (c.vgpr) hipLaunchKernelGGL((pat_kernel<INFLIGHT, 1, 0>), dim3(256), dim3(threads), lds, 0, src, bytes, c, sink);
        else if (c.aux) hipLaunchKernelGGL((pat_kernel<INFLIGHT, 0, 16>), dim3(256), dim3(threads), lds, 0, src, bytes, c, sink);
        else hipLaunchKernelGGL((pat_kernel<INFLIGHT, 0, 0>), dim3(256), dim3(threads), lds, 0, src, bytes, c, sink);
    };
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 256.0 * c.loaders * c.iters * INFLIGHT * 1024.0 / (ms * 1e-3) / 1e9 / 256.0;      // GB/s per CU
}

int main() {
    const unsigned bytes = 4u << 20;
    char *src; float *sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 64);
    printf("# one workgroup per CU, 8 instructions (8 KB) in flight per loader wave, 2 MB L2-resident region; GB/s per CU\n");
    printf("%-34s %8s %8s %8s %8s\n", "pattern", "2ld+4mf", "4ld+4mf", "2ld", "4ld");
    struct Row { const char *name; int rows, stride, swz, shared, vgpr, aux; };
    const Row rows[] = {
        {"contiguous 1KB, private", 0, 0, 0, 0, 0, 0},
        {"contiguous 1KB, shared", 0, 0, 0, 1, 0, 0},
        {"contiguous 1KB, private, ->VGPR", 0, 0, 0, 0, 1, 0},
        {"contiguous 1KB, shared, ->VGPR", 0, 0, 0, 1, 1, 0},
        {"8x128B stride 512, private", 1, 512, 0, 0, 0, 0},
        {"8x128B stride 512, xor7", 1, 512, 1, 0, 0, 0},
        {"8x128B stride 512, xor-half", 1, 512, 2, 0, 0, 0},
        {"8x128B stride 512, rotate", 1, 512, 3, 0, 0, 0},
        {"8x128B stride 512, xor7, shared", 1, 512, 1, 1, 0, 0},
        {"8x128B stride 512, xor7, sc1", 1, 512, 1, 0, 0, 16},
        {"8x128B stride 512, xor7, ->VGPR", 1, 512, 1, 0, 1, 0},
        {"8x128B stride 512, ->VGPR", 1, 512, 0, 0, 1, 0},
        {"8x128B stride 512, ->VGPR->ds_write", 1, 512, 0, 0, 2, 0},
        {"contiguous 1KB, ->VGPR->ds_write", 0, 0, 0, 0, 2, 0},
        {"8x128B stride 2048, private", 1, 2048, 0, 0, 0, 0},
        {"8x128B stride 2048, xor7", 1, 2048, 1, 0, 0, 0},
        {"8x128B stride 2048, xor7, shared", 1, 2048, 1, 1, 0, 0},
        {"8x128B stride 128 (dense), xor7", 1, 128, 1, 0, 0, 0},
        {"8x128B stride 256, xor7", 1, 256, 1, 0, 0, 0},
    };
    const int iters = 300;
    for (const Row &r : rows) {
        double v[4];
        const int ld[4] = {2, 4, 2, 4}, mf[4] = {4, 4, 0, 0};
        for (int k = 0; k < 4; ++k) {
            Cfg c{r.rows, r.stride, r.swz, r.shared, r.vgpr, r.aux, ld[k], mf[k], iters};
            v[k] = run(src, bytes, c, sink);
        }
        printf("%-34s %8.1f %8.1f %8.1f %8.1f\n", r.name, v[0], v[1], v[2], v[3]);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
