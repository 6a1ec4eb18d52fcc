// kloop_skeleton.hip -- measurement aid (not part of the library).  Round 3.
//
// profiles/r03_seq_probe2_per_cu_bound.txt: the K loop of conv_wreg / conv_seq costs ~0.36 us per 64x128 K tile whatever the
// other CUs do, and neither operand stream alone nor the MFMAs alone set that time.  profiles/r02_dma_patterns.txt: ONE wave
// sustains a fixed ~14-17 GB/s of contiguous loads (~8 GB/s of 8-rows-x-128-B gathers) whatever it has in flight.  In the
// kernel 4 consumer waves pull the weights (16 KB per K tile -> 4 KB per wave) and 4 producer waves the activation rows
// (8 KB -> 2 KB per wave): both streams sit at the per-wave ceiling.  Round 2 doubled each side ALONE (eight producers; eight
// consumers) and saw nothing -- the other side was still the floor.  This skeleton has the K loop's traffic and
// synchronisation and nothing else, so the wave split can be swept in seconds:
//
//   consumers (NC waves): weight fragments global -> VGPR (1 KB contiguous per instruction, LOOK tiles ahead), A fragments
//                         from LDS (ds_read_b128), MFMAs, one s_barrier per K tile
//   producers (NP waves): activation rows global -> LDS by LDS-DMA (8 rows x 128 B per instruction, LOOK tiles ahead),
//                         counted vmcnt, the same s_barrier
//   tile (BM x BN): W bytes per K tile = BN * 128, A bytes = BM * 128, MFMAs (32x32x16) = BM/32 * BN/32 * 4
//
// one workgroup per CU, every CU streams private 2 MB regions that stay L2-resident.  Prints us per K tile and the GB/s per
// CU that implies.  Build: hipcc -O3 --offload-arch=gfx950 tools/kloop_skeleton.hip -o /tmp/kloop_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

struct Cfg { int nc, np, bm, bn, tiles, look; };

// WPT = weight instructions per consumer wave per K tile, APT = activation instructions per producer wave per K tile,
// MPT = MFMAs per consumer wave per K tile, LOOK = tiles in flight
template <int WPT, int APT, int MPT, int LOOK, int MAXT, int full_a>
__global__ __launch_bounds__(MAXT) void kloop(const char *src, unsigned src_bytes, int nc, int np, int tiles, float *sink, int wshare) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, src_bytes, 0x00020000);
    const unsigned region = 2048u * 1024u;
    if (wave < nc) {
        // ---- consumer: W fragments LOOK tiles ahead in a register ring, MFMAs on them, A fragments from LDS
        // (wshare: every workgroup streams the SAME weight region, like the CUs of a layer do)
        unsigned off = ((((wshare ? 0 : blockIdx.x * 16) + wave) * 4099u) % 2048u) * 1024u + lane * 16;
        auto next = [&]() { const unsigned o = off; off = (off + 64 * 1024) & (region - 1); return (int)o; };
        uint4v w[LOOK][WPT];
#pragma unroll
        for (int t = 0; t < LOOK; ++t)
#pragma unroll
            for (int j = 0; j < WPT; ++j) w[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, next(), 0, 0);
        floatx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = (float)(lane + i);
        const unsigned char *la = smem + (lane & 31) * 128 + (lane >> 5) * 16;
        __builtin_amdgcn_s_barrier();
        for (int t0 = 0; t0 < tiles; t0 += LOOK) {
#pragma unroll
            for (int t = 0; t < LOOK; ++t) {
                if constexpr (!full_a) {
                    half8 a0 = *(const half8 *)(la + ((t0 + t) & 3) * 8192);
                    half8 a1 = *(const half8 *)(la + ((t0 + t) & 3) * 8192 + 4096);
#pragma unroll
                    for (int m = 0; m < MPT; ++m) {
                        const half8 b = __builtin_bit_cast(half8, w[t][m % WPT]);
                        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16((m & 1) ? a1 : a0, b, acc[m & 3], 0, 0, 0);
                    }
                } else {
                    // the real kernel's LDS traffic: one A fragment (1 KB wave read) per TWO MFMAs (a wave's 64 columns = two
                    // 32-column blocks), fragments fetched two MFMA pairs ahead
                    half8 af[MPT / 2];
#pragma unroll
                    for (int m = 0; m < MPT / 2; ++m) {       // row block m / 4, k-step m % 4; the kernel's 16-byte-slot swizzle (conflict-free)
                        const int row = (m >> 2) * 32 + (lane & 31), slot = ((m & 3) * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
                        af[m] = *(const half8 *)(smem + ((t0 + t) & 3) * 8192 + ((row * 128) & 8191) + slot * 16);
                    }
#pragma unroll
                    for (int m = 0; m < MPT; ++m) {
                        const half8 b = __builtin_bit_cast(half8, w[t][m % WPT]);
                        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m >> 1], b, acc[m & 3], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < WPT; ++j) w[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, next(), 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 123.456f) sink[0] = s;
    } else {
        // ---- producer: LDS-DMA of 8 rows x 128 B per instruction, LOOK tiles ahead, counted vmcnt, one barrier per tile
        const int pw = wave - nc;
        const unsigned who = blockIdx.x * 16 + wave;
        const int r8 = lane >> 3, slot = lane & 7;
        const unsigned stride = 2048u, nrows = region / stride;
        unsigned row0 = ((who * 37u) % (nrows / 64u)) * 64u, kt = 0;
        auto issue = [&](int t) {
#pragma unroll
            for (int j = 0; j < APT; ++j) {
                const unsigned row = row0 + (unsigned)((j * 8 + r8) & 63);
                const int s = (slot ^ (int)((row >> 1) & 7u)) & 7;
                const unsigned o = (row * stride + kt * 128u + (unsigned)s * 16u) & (region - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + (t & 3) * 8192 + ((pw * APT + j) & 7) * 1024), 16,
                                                         (int)o, 0, 0, 16);
            }
            if (++kt == stride / 128u) { kt = 0; row0 = (row0 + 64u) % nrows; }
        };
#pragma unroll
        for (int t = 0; t < LOOK; ++t) issue(t);
        for (int t = 0; t < tiles; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LOOK - 1) * APT) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(t + LOOK);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // pairs with the consumers' prologue barrier
    }
}

template <int WPT, int APT, int MPT, int LOOK, int FA>
static double time_it(const char *src, unsigned bytes, const Cfg &c, float *sink, int wshare) {
    const int threads = 64 * (c.nc + c.np);
    // registers: the weight ring + 64 accumulators must fit the per-wave budget (256 with <= 8 waves, 128 with 16)
    if (WPT * LOOK * 4 + 64 + 24 > (threads <= 512 ? 250 : 126)) return 0.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (threads <= 512) hipLaunchKernelGGL((kloop<WPT, APT, MPT, LOOK, 512, FA>), dim3(256), dim3(threads), 40 * 1024, 0, src, bytes, c.nc, c.np, c.tiles, sink, wshare);
        else hipLaunchKernelGGL((kloop<WPT, APT, MPT, LOOK, 1024, FA>), dim3(256), dim3(threads), 40 * 1024, 0, src, bytes, c.nc, c.np, c.tiles, sink, wshare);
    };
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / c.tiles;                              // us per K tile
}

#define CASE(WPT_, APT_, MPT_)                                                                            \
    if (wpt == WPT_ && apt == APT_ && mpt == MPT_) {                                                      \
        us2 = full_a ? time_it<WPT_, APT_, MPT_, 2, 1>(src, bytes, c, sink, wshare) : time_it<WPT_, APT_, MPT_, 2, 0>(src, bytes, c, sink, wshare); \
        us4 = full_a ? time_it<WPT_, APT_, MPT_, 4, 1>(src, bytes, c, sink, wshare) : time_it<WPT_, APT_, MPT_, 4, 0>(src, bytes, c, sink, wshare); \
        done = true;                                                                                      \
    }

int main() {
    const unsigned bytes = 4u << 20;
    char *src; float *sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 64);
    printf("# K-loop skeleton, one workgroup per CU, 256 workgroups; us per K tile (64 halves of K) with the operands 2 / 4 tiles ahead\n");
    printf("%-9s %3s %3s | W/wave A/wave MFMA/wave |  look2 us   GB/s/CU  MFMA%% |  look4 us   GB/s/CU  MFMA%%\n", "tile", "NC", "NP");
    const int shapes[4][2] = {{64, 128}, {64, 256}, {128, 128}, {128, 256}};
    const int splits[6][2] = {{4, 4}, {4, 8}, {8, 4}, {8, 8}, {4, 12}, {8, 12}};
    for (int wshare = 0; wshare < 2; ++wshare)
    for (int full_a = 0; full_a < 2; ++full_a)
    for (auto &sh : shapes)
        for (auto &sp : splits) {
            Cfg c{sp[0], sp[1], sh[0], sh[1], 2000, 2};
            const int wkb = sh[1] * 128 / 1024, akb = sh[0] * 128 / 1024, mf = (sh[0] / 32) * (sh[1] / 32) * 4;
            if (wkb % c.nc || mf % c.nc) continue;
            const int wpt = wkb / c.nc, mpt = mf / c.nc;
            int apt = (akb + c.np - 1) / c.np;             // (12 producers on 8 KB: 1 instruction each, four of them idle -> skip)
            if (akb % c.np) continue;
            double us2 = 0, us4 = 0;
            bool done = false;
            CASE(2, 1, 4) CASE(2, 2, 4) CASE(2, 2, 8) CASE(2, 4, 8) CASE(4, 1, 8) CASE(4, 2, 8) CASE(4, 2, 16) CASE(4, 4, 16)
            CASE(8, 1, 16) CASE(8, 2, 16) CASE(8, 2, 32) CASE(8, 4, 32)
            if (!done) { printf("%3dx%-5d %3d %3d | (no instantiation for W %d A %d MFMA %d)\n", sh[0], sh[1], c.nc, c.np, wpt, apt, mpt); continue; }
            const double kb = wkb + akb, ideal = mf * 32.0 / 4.0 / 2.4e3;      // MFMA time of the tile on 4 SIMDs at 2.4 GHz, us
            printf("%-12s %3dx%-5d %3d %3d | %5d %6d %9d | %8.3f %9.1f %6.1f | %8.3f %9.1f %6.1f\n", wshare ? (full_a ? "fullA/sameW" : "2frag/sameW") : (full_a ? "fullA" : "2frag"), sh[0], sh[1], c.nc, c.np, wpt, apt, mpt, us2,
                   us2 > 0 ? kb * 1024 / us2 / 1e3 : 0.0, us2 > 0 ? 100.0 * ideal / us2 : 0.0, us4, us4 > 0 ? kb * 1024 / us4 / 1e3 : 0.0,
                   us4 > 0 ? 100.0 * ideal / us4 : 0.0);
            fflush(stdout);
        }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
