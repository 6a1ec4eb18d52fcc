// l1_block.hip -- one ResNet layer1 Bottleneck (experiments/siammask_sharp/resnet.py:64-103,159) as ONE launch (fp16):
//   conv1 1x1 (Cin -> 64) + BN + ReLU -> conv2 3x3 p1 (64 -> 64) + BN + ReLU -> conv3 1x1 (64 -> 256) + BN
//   (+ shortcut: the block input, or for block 0 the 1x1 projection 64 -> 256 + BN) -> ReLU.
//
// Layer1 works on the 63 x 63 (search) / 31 x 31 (template) maps with 64 / 256 channels: 1 % of the network's FLOPs but
// 16 MB tensors per layer at B = 8, three launches per block at 9-16 us each -- 100 us of the 0.68 ms step (15 %), 2.5-3x its
// memory floor, and none of it arithmetic.  The round-2 attempt at fusing a block streamed the weights and ran one workgroup
// per CU with a barrier per chunk: slower.  What is different here:
//   * the WHOLE block's weights (16 + 36 + 16 KB) live in the four waves' REGISTERS for the lifetime of the workgroup
//     (v_mfma_f32_16x16x32_f16 fragments: wave w owns output channels 16w..16w+15 of conv1 / conv2 and 64w..64w+63 of
//     conv3): no weight traffic and no barrier inside a convolution;
//   * a workgroup = one 8 x 8 output tile; the 10 x 10 halo of the block input is the ONLY activation read from memory
//     (it also serves the residual add), conv1's and conv2's outputs stay in LDS as fp16 (exactly the rounding they get
//     when they go through HBM), the 64 x 256 result leaves as full 512-byte pixel rows;
//   * 76 KB of LDS and <= 256 VGPRs: two workgroups per CU, 512 of them at B = 8 -- the load of one tile overlaps the
//     MFMAs of the other; four barriers per tile.
// conv2's zero padding applies to conv1's OUTPUT: halo pixels outside the image are written as zeros, not relu(bias).
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int LB_T = 8;                       // output tile side
constexpr int LB_H = LB_T + 2;                // halo side (10)
constexpr int LB_NH = LB_H * LB_H;            // halo pixels (100)
constexpr int LB_NO = LB_T * LB_T;            // output pixels (64)
constexpr int LB_TP = 64 * 2 + 16;            // LDS pitch of a 64-channel pixel row (144 B: conflict-free 16-row fragments)

template <int CIN> struct LbLds {
    static constexpr int XP = CIN * 2 + 16;                                  // pitch of a block-input pixel row
    static constexpr int XS = LB_NH * XP;
    static constexpr int T1 = LB_NH * LB_TP, T2 = LB_NO * LB_TP;
    static constexpr int YP = 256 * 2 + 16;
    static constexpr int YS = CIN == 256 ? 0 : LB_NO * YP;                   // block 0 stages its output separately
    static constexpr int TOTAL = XS + T1 + T2 + YS;
};

template <int CIN>
__global__ __launch_bounds__(256, 2) void l1_block_kernel(const L1BlockParams p) {
    constexpr bool FIRST = CIN == 64;
    typedef LbLds<CIN> L;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::TOTAL];
    unsigned char *xs = smem, *t1 = smem + L::XS, *t2 = t1 + L::T1, *ys = t2 + L::T2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    set_wave_prio(p.prio);                                    // (smk_tune main_prio, see stem_pool.hip)
    const int fr = lane & 15, kq = lane >> 4;
    const int tpr = (p.S + LB_T - 1) / LB_T;
    const int b = blockIdx.x / (tpr * tpr);
    const int tt = blockIdx.x - b * (tpr * tpr);
    const int y0 = (tt / tpr) * LB_T, x0 = (tt - (tt / tpr) * tpr) * LB_T;
    // (SMK_L1_CLK=1, eager runs: 100 MHz stamps of workgroup 0 at the phase boundaries -- measurement aid)
    const bool clk = p.clk && blockIdx.x == 0 && tid == 0;
    if (clk) p.clk[0] = wall_clock64();

    // ---- the block's weights: fragments of v_mfma_f32_16x16x32_f16 in fragment order (w_frag16: one contiguous KB per (16 output
    // channels, 32 k) operand, lane = (n % 16) + 16 * ((k % 32) / 8)), every load a fully coalesced 1 KB wave instruction.
    // The weights are the A operand and the activations the B operand (C^T = W x A^T), so that a lane ends up with FOUR
    // CONSECUTIVE CHANNELS of one pixel -- 8-byte LDS accesses in the epilogues instead of four 2-byte ones.
    constexpr int KS1 = CIN / 32;
    const uint4v *w1 = (const uint4v *)p.w1, *w2 = (const uint4v *)p.w2, *w3 = (const uint4v *)p.w3, *wd = (const uint4v *)p.wd;
    const int k1s = p.K1pad >> 5, k2s = p.K2pad >> 5, k3s = p.K3pad >> 5, kds = p.Kdpad >> 5;
    half8 w1f[KS1], w2f[18], w3f[4][2], wdf[FIRST ? 4 : 1][2];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) w1f[ks] = __builtin_bit_cast(half8, w1[((size_t)wave * k1s + ks) * 64 + lane]);
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) w2f[ks] = __builtin_bit_cast(half8, w2[((size_t)wave * k2s + ks) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            w3f[j][ks] = __builtin_bit_cast(half8, w3[((size_t)(4 * wave + j) * k3s + ks) * 64 + lane]);
            if constexpr (FIRST) wdf[j][ks] = __builtin_bit_cast(half8, wd[((size_t)(4 * wave + j) * kds + ks) * 64 + lane]);
        }
    // biases of this lane's four channels per 16-channel block (channel = block * 16 + 4 kq + i)
    float b1v[4], b2v[4], b3v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b1v[i] = p.b1[16 * wave + 4 * kq + i];
        b2v[i] = p.b2[16 * wave + 4 * kq + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b3v[j][i] = p.b3[(4 * wave + j) * 16 + 4 * kq + i];
            if constexpr (FIRST) b3v[j][i] += p.bd[(4 * wave + j) * 16 + 4 * kq + i];
        }
    }

    // ---- the 10 x 10 halo of the block input -> LDS (pixels outside the image: zeros) -----------------------------------------
    // All of a thread's 16-byte loads are issued before the first LDS write: with two workgroups per CU there is no third
    // one to hide a load -> wait -> ds_write chain of 13 memory latencies behind.  (Unconditional loads of clamped
    // coordinates + a select: a load under a branch is waited for at the end of that branch.)
    {
        constexpr int VPP = CIN / 8;                          // 16-byte vectors per pixel
        constexpr int NLD = (LB_NH * VPP + 255) / 256;        // 13 (CIN = 256) / 4 (CIN = 64)
        const _Float16 *x = (const _Float16 *)p.x + (size_t)b * p.S * p.S * CIN;
        uint4v val[NLD];
        bool ok[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int v = tid + i * 256;
            v = v < LB_NH * VPP ? v : tid;                    // (beyond the tile: re-load the own first vector)
            const int px = v / VPP, q = v - px * VPP;
            const int r = px / LB_H, c = px - r * LB_H;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            ok[i] = (unsigned)gy < (unsigned)p.S && (unsigned)gx < (unsigned)p.S;
            const int cy = min(max(gy, 0), p.S - 1), cx = min(max(gx, 0), p.S - 1);
            val[i] = *(const uint4v *)(x + ((size_t)cy * p.S + cx) * CIN + q * 8);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * 256;
            if (v < LB_NH * VPP) {
                const int px = v / VPP, q = v - px * VPP;
                *(uint4v *)(xs + px * L::XP + q * 16) = ok[i] ? val[i] : uint4v{0u, 0u, 0u, 0u};
            }
        }
    }
    __syncthreads();
    if (clk) p.clk[1] = wall_clock64();

    // ---- conv1: 100 halo pixels (7 row blocks of 16, in pairs: two independent accumulator chains) x this wave's 16 channels ------
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    auto c1_store = [&](int mt, const floatx4 &acc) {
        const int px = mt * 16 + fr;                           // this lane's pixel of the block (the MFMA column)
        if (px < LB_NH) {
            const int r = px / LB_H, c = px - r * LB_H;
            const bool in_img = (unsigned)(y0 - 1 + r) < (unsigned)p.S && (unsigned)(x0 - 1 + c) < (unsigned)p.S;
            half4 h;
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = (_Float16)(in_img ? fmaxf(acc[i] + b1v[i], 0.f) : 0.f);
            *(half4 *)(t1 + px * LB_TP + (16 * wave + 4 * kq) * 2) = h;
        }
    };
    // (row blocks FOUR at a time: four independent accumulator chains and four LDS fragment reads in flight per k-step -- with one
    //  wave of the workgroup per SIMD nothing else hides the ds_read latency, which is what a phase of this kernel costs)
    for (int mt = 0; mt < 8; mt += 4) {
        const unsigned char *a[4];
        floatx4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int m = (mt + u) * 16 + fr;
            m = m < LB_NH ? m : LB_NH - 1;                     // (the eighth block does not exist: computed on clamped rows, not stored)
            a[u] = xs + m * L::XP + kq * 16;
            acc[u] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            half8 f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = *(const half8 *)(a[u] + ks * 64);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[ks], f[u], acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mt + u < 7) c1_store(mt + u, acc[u]);
    }
    __syncthreads();
    if (clk) p.clk[2] = wall_clock64();

    // ---- conv2: 3 x 3 over the halo image, 64 output pixels (4 row blocks, in pairs) x this wave's 16 channels, K = 9 x 64 ---------
    {
        const unsigned char *a[4];
        floatx4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = u * 16 + fr;
            a[u] = t1 + ((m >> 3) * LB_H + (m & 7)) * LB_TP + kq * 16;
            acc[u] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            const int tap = ks >> 1, dy = tap / 3, dx = tap - 3 * dy;
            const int o = (dy * LB_H + dx) * LB_TP + (ks & 1) * 64;
            half8 f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = *(const half8 *)(a[u] + o);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[ks], f[u], acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            half4 h;
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = (_Float16)fmaxf(acc[u][i] + b2v[i], 0.f);
            *(half4 *)(t2 + (u * 16 + fr) * LB_TP + (16 * wave + 4 * kq) * 2) = h;
        }
    }
    __syncthreads();
    if (clk) p.clk[3] = wall_clock64();

    // ---- conv3 (+ the 1x1 projection shortcut of block 0): 64 pixels x this wave's 64 channels, K = 64; + residual; ReLU -----------
    for (int mt = 0; mt < 4; ++mt) {
        const int m = mt * 16 + fr;
        const unsigned char *a0 = t2 + m * LB_TP + kq * 16;
        const int cm = ((m >> 3) + 1) * LB_H + (m & 7) + 1;       // the same pixel in the halo image
        floatx4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 a = *(const half8 *)(a0 + ks * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[j][ks], a, acc[j], 0, 0, 0);
        }
        if constexpr (FIRST) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 a = *(const half8 *)(xs + cm * L::XP + kq * 16 + ks * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdf[j][ks], a, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = (4 * wave + j) * 16 + 4 * kq;          // this lane's four channels of the block
            half4 h;
            if constexpr (FIRST) {
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)fmaxf(acc[j][i] + b3v[j][i], 0.f);
                *(half4 *)(ys + m * L::YP + n0 * 2) = h;
            } else {
                half4 *slot = (half4 *)(xs + cm * L::XP + n0 * 2);    // residual in, result out: same lane, same place
                const half4 r = *slot;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)fmaxf(acc[j][i] + b3v[j][i] + (float)r[i], 0.f);
                *slot = h;
            }
        }
    }
    __syncthreads();
    if (clk) p.clk[4] = wall_clock64();

    // ---- the 64 x 256 result: full 512-byte pixel rows to HBM ------------------------------------------------------------------
    _Float16 *y = (_Float16 *)p.y + (size_t)b * p.S * p.S * 256;
    for (int v = tid; v < LB_NO * 32; v += 256) {
        const int px = v >> 5, q = v & 31;
        const int gy = y0 + (px >> 3), gx = x0 + (px & 7);
        if (gy < p.S && gx < p.S) {
            const unsigned char *src = FIRST ? ys + px * L::YP + q * 16 : xs + (((px >> 3) + 1) * LB_H + (px & 7) + 1) * L::XP + q * 16;
            *(uint4v *)(y + ((size_t)gy * p.S + gx) * 256 + q * 8) = *(const uint4v *)src;
        }
    }
    if (clk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.clk[5] = wall_clock64();
    }
}

int launch_l1_block(const L1BlockParams &p, void *stream) {
    if (!p.x || !p.y || !p.w1 || !p.w2 || !p.w3 || (p.Cin != 64 && p.Cin != 256) || (p.Cin == 64 && !p.wd)) return -1;
    if (p.K1pad < p.Cin || p.K2pad < 576 || p.K3pad < 64 || (p.K1pad | p.K2pad | p.K3pad | p.Kdpad) % 32) return -1;
    const int tpr = (p.S + LB_T - 1) / LB_T;
    const dim3 grid(p.B * tpr * tpr), block(256);
    // smk_tune "front_occ1" bit 0 (A/B knob of MEASURE builds): unused dynamic LDS on top of the static 72 / 76 KB, so that ONE workgroup owns
    // a CU and the other half of its LDS and registers stays free for another stream's kernels (the pipelined step's tail).  Measured:
    // +3.5 % per step -- the tail does not use what is freed (profiles/r06g_front_occupancy_ab.txt).
#ifdef SMK_MEASURE
    const unsigned pad = (g_tune.front_occ1 & 1) ? (p.Cin == 64 ? 12288u : 8192u) : 0u;
#else
    const unsigned pad = 0u;
#endif
    if (p.Cin == 64) hipLaunchKernelGGL(l1_block_kernel<64>, grid, block, pad, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(l1_block_kernel<256>, grid, block, pad, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
