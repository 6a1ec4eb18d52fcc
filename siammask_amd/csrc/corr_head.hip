// corr_head.hip -- depth-wise cross-correlation + head.0 (1x1 + BN + ReLU) + head.3 of the cls / loc branches as ONE launch (fp16).
//   models/rpn.py:32-38 (conv2d_dw_group), :50-60 (DepthCorr.head), :62-72 (forward_corr + head); experiments/siammask_sharp/
//   custom.py:86-99 (the mask branch's DepthCorr: its head.3 -- 256 -> 63*63 -- stays with the Refine chain launch).
//
// Why (round 4): behind conv_search the B = 8 step ran dw_xcorr (15 us) -> head.0 (11 us) -> cls.head.3 + loc.head.3 (8 us) as
// three dependent launches at the launch floor, each handing a 2.4 MB tensor to the next through memory; none of them is
// bandwidth- or matrix-bound.  A 5-row band of one (stream, branch) is 125 output pixels = one 128-row GEMM tile, and head.0
// needs all 256 channels of a pixel and nothing of its neighbours -- so the band's correlation output stays in LDS and is the
// activation operand of head.0, whose output stays in LDS and is the operand of head.3:
//
//   workgroup  = (band of 5 output rows, branch, stream): 5 x 3 x B workgroups of 10 waves, 144 KB of LDS
//   phase 1    : the correlation in two rounds of 128 channels (9 x 29 input pixels x 128 channels staged per round, the second
//                round's loads in flight in registers while the first one computes); the inner loop is dw_xcorr_kernel's -- a
//                thread owns a channel pair and a half-row strip and slides the 5-tap window in registers, fp32 fmaf in the same
//                order -- so `corr` is bit-identical to the stand-alone kernel's; it is written to LDS as fp16 (the rounding the
//                tensor gets in memory) and from there to memory by two spare waves while ...
//   phase 2    : ... waves 0-7 run head.0 as W x A^T (wave = 32 output channels x the 4 row fragments; the whole 16-k-step weight
//                panel of a wave, 16 KB, was requested at kernel entry and has arrived long before); bias + ReLU -> fp16 -> LDS
//   phase 3    : the head.0 tile goes to memory in full lines (the mask head and the tests read it); on the cls / loc branches four
//                waves run head.3 (10 / 20 output channels inside one 32-channel block) on it and write NCHW fp32 directly:
//                a lane holds one pixel, so the 32 lanes of a register write 128 contiguous bytes of one channel plane.
// fp16 only (the fp32 path keeps the three launches); per-stream template taps `zk` are the cached conv_kernel(zf).
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int CH_W = 29, CH_K = 5, CH_WO = 25, CH_BR = 5;          // search map, taps, output map, output rows per band
constexpr int CH_RIN = CH_BR + CH_K - 1;                           // 9 input rows per band
constexpr int CH_PIX = CH_BR * CH_WO;                              // 125 output pixels per band
constexpr int CH_NT = 640;                                         // 10 waves
constexpr int CH_SX = CH_RIN * CH_W * 128 * 2;                     // [261 pixels][128 channels] f16 of one round
constexpr int CH_SK = CH_K * CH_K * 256 * 2;                       // [25 taps][256 channels]
constexpr int CH_CP = 528;                                         // row pitch of the 128 x 256 f16 tiles: 16 rows hit 16 different bank quads
constexpr int CH_CO = 128 * CH_CP;
constexpr int CH_SW = 13;                                          // outputs per thread strip (half of a 25-wide row)
constexpr int CH_NV = CH_RIN * CH_W * 16;                          // 16-byte vectors of one round (4176)
constexpr int CH_NLD = (CH_NV + CH_NT - 1) / CH_NT;                // ... per thread (7)
static_assert(CH_SX + CH_SK >= CH_CO, "the head.0 tile re-uses the staging area");
static_assert(CH_SX + CH_SK + CH_CO <= 160 * 1024, "LDS");

__global__ __launch_bounds__(CH_NT) void corr_head_kernel(const CorrHeadParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SX + CH_SK + CH_CO];
    _Float16 *sx = (_Float16 *)smem;
    _Float16 *sk = (_Float16 *)(smem + CH_SX);
    unsigned char *co = smem + CH_SX + CH_SK;                       // corr tile [128][CH_CP]
    unsigned char *ht = smem;                                       // head.0 tile [128][CH_CP] (phase 2 on: sx / sk are dead)
    const int band = blockIdx.x, br = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fm = lane & 31, fh = lane >> 5;
    const int Cs = p.Cs;
    const size_t opix0 = (size_t)b * (CH_WO * CH_WO) + band * CH_PIX;      // first output pixel of the band
    // pipelined frame step, depth 2: this launch STARTING means conv_search has drained -- the first workgroup tells the previous frame's
    // Refine chain + mask head launch (its gate polls this semaphore) that the chip's idle CUs are its own from here on
    if (p.start_sem && tid == 0 && band == 0 && br == 0 && b == 0) __hip_atomic_fetch_add(p.start_sem, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- phase 1: the correlation, two rounds of 128 channels -------------------------------------------------------------------
    const _Float16 *xs = p.xs + ((size_t)(b * CH_W + band * CH_BR) * CH_W) * Cs + br * 256;      // the band's 261 input pixels are contiguous
    for (int v = tid; v < CH_K * CH_K * 32; v += CH_NT) {             // taps [25][256]
        const int tap = v >> 5, q = v & 31;
        *(uint4v *)(sk + tap * 256 + q * 8) = *(const uint4v *)(p.zk + ((size_t)b * (CH_K * CH_K) + tap) * Cs + br * 256 + q * 8);
    }
    uint4v rg[CH_NLD];
    auto load_round = [&](int r) {
#pragma unroll
        for (int i = 0; i < CH_NLD; ++i) {
            int v = tid + i * CH_NT;
            if (v >= CH_NV) v = tid;                                  // (the tail re-reads a valid vector: no branch between load and use)
            const int pix = v >> 4, q = v & 15;
            rg[i] = *(const uint4v *)(xs + (size_t)pix * Cs + r * 128 + q * 8);
        }
    };
    auto store_round = [&]() {
#pragma unroll
        for (int i = 0; i < CH_NLD; ++i) {
            const int v = tid + i * CH_NT;
            if (v < CH_NV) *(uint4v *)(sx + (v >> 4) * 128 + (v & 15) * 8) = rg[i];
        }
    };
    const int g = tid >= 320 ? 1 : 0, t = tid - g * 320;              // two groups of 320 threads: 64 channels each per round
    const int cp = t & 31, strip = t >> 5;                            // channel pair, (row, half) strip
    const int ri = strip >> 1, hf = strip & 1;
    const int j0 = hf ? CH_SW : 0, jn = hf ? CH_WO - CH_SW : CH_SW;   // 13 + 12 outputs
    auto xcorr_round = [&](int r) {
        const int cch = r * 128 + g * 64 + cp * 2;                    // this thread's channel pair inside the branch
        floatx2 acc[CH_SW];
#pragma unroll
        for (int j = 0; j < CH_SW; ++j) acc[j] = floatx2{0.f, 0.f};
#pragma unroll 1
        for (int u = 0; u < CH_K; ++u) {
            floatx2 tap[CH_K];
#pragma unroll
            for (int v = 0; v < CH_K; ++v) {
                const half2_t h = *(const half2_t *)(sk + (u * CH_K + v) * 256 + cch);
                tap[v] = floatx2{(float)h[0], (float)h[1]};
            }
            const _Float16 *srow = sx + ((ri + u) * CH_W + j0) * 128 + g * 64 + cp * 2;
#pragma unroll
            for (int tt = 0; tt < CH_SW + CH_K - 1; ++tt) {
                // input column j0 + tt contributes to outputs jj = tt - v, v = 0..4 (dw_xcorr_kernel's order: bit-identical sums)
                floatx2 xv = floatx2{0.f, 0.f};
                if (tt < jn + CH_K - 1) {
                    const half2_t h = *(const half2_t *)(srow + tt * 128);
                    xv = floatx2{(float)h[0], (float)h[1]};
                }
#pragma unroll
                for (int v = 0; v < CH_K; ++v) {
                    const int jj = tt - v;
                    if (jj >= 0 && jj < CH_SW) {
                        acc[jj][0] = fmaf(xv[0], tap[v][0], acc[jj][0]);
                        acc[jj][1] = fmaf(xv[1], tap[v][1], acc[jj][1]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CH_SW; ++j)
            if (j < jn) {
                const half2_t h = {(_Float16)acc[j][0], (_Float16)acc[j][1]};
                *(half2_t *)(co + (ri * CH_WO + j0 + j) * CH_CP + cch * 2) = h;
            }
    };
    load_round(0);
    store_round();
    load_round(1);                                                    // in flight while round 0 computes
    __syncthreads();
    xcorr_round(0);
    __syncthreads();                                                  // everybody is done with round 0's sx
    store_round();
    // head.0's weight panel of this wave -- 32 output channels x K = 256 = 16 fragments of 1 KB -- requested here (the staging
    // registers are free now): it arrives while round 1 computes
    half8 wf[16];
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.w0_frag, 0, p.w0_bytes, 0x00020000);
    const int wblk = (br * 8 + (wave & 7)) * 16;                      // 32-row block of the grouped pack (group = branch), in fragments
    if (wave < 8) {
#pragma unroll
        for (int s = 0; s < 8; ++s)                                   // (half of the panel: all of it beside round 1's registers spills)
            wf[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, ((wblk + s) * 64 + lane) * 16, 0, 0));
    }
    __syncthreads();
    xcorr_round(1);
    __syncthreads();                                                  // the corr tile is complete

    // ---- phase 2: head.0 on waves 0-7 (W x A^T: channels x pixels); waves 8-9 write the corr tile to memory ---------------------
    floatx4 bq[4];
    if (wave < 8) {
#pragma unroll
        for (int s = 8; s < 16; ++s)
            wf[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, ((wblk + s) * 64 + lane) * 16, 0, 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *(const floatx4 *)(p.b0 + br * 256 + wave * 32 + 8 * q + 4 * fh);
    }
    // (two row fragments at a time against the whole register-resident panel: 10 waves leave 168 registers per lane, and
    //  four accumulators + the panel + the operand fragments do not fit; the head.0 tile overwrites the staging area, which
    //  nobody reads any more, so each pair of fragments is written out as soon as it is done)
    if (wave < 8) {
#pragma unroll
        for (int f0 = 0; f0 < 4; f0 += 2) {
            floatx16 acc0[2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[f][r] = bq[r >> 2][r & 3];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                half8 a[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) a[f] = *(const half8 *)(co + ((f0 + f) * 32 + fm) * CH_CP + s * 32 + fh * 16);
#pragma unroll
                for (int f = 0; f < 2; ++f) acc0[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], a[f], acc0[f], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 o;
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] = (_Float16)fmaxf(acc0[f][4 * q + u], 0.f);
                    *(half4 *)(ht + ((f0 + f) * 32 + fm) * CH_CP + (wave * 32 + 8 * q + 4 * fh) * 2) = o;
                }
        }
    } else {
        _Float16 *corr = p.corr + opix0 * Cs + br * 256;
        for (int v = tid - 512; v < CH_PIX * 32; v += 128) {
            const int row = v >> 5, q = v & 31;
            *(uint4v *)(corr + (size_t)row * Cs + q * 8) = *(const uint4v *)(co + row * CH_CP + q * 16);
        }
    }
    // head.3's weight fragments (cls / loc branches: block 0 of the pack holds the 10 / 20 real rows) into the panel's registers
    const bool has3 = br < 2 && p.w3_frag[br] != nullptr;
    if (has3 && wave < 4) {
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w3_frag[br], 0, p.w3_bytes[br], 0x00020000);
#pragma unroll
        for (int s = 0; s < 16; ++s) wf[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, (s * 64 + lane) * 16, 0, 0));
    }
    __syncthreads();

    // ---- phase 3: the head.0 tile to memory; head.3 of the cls / loc branches ----------------------------------------------------
    {
        _Float16 *h0 = p.h0 + opix0 * Cs + br * 256;
        for (int v = tid; v < CH_PIX * 32; v += CH_NT) {
            const int row = v >> 5, q = v & 31;
            *(uint4v *)(h0 + (size_t)row * Cs + q * 8) = *(const uint4v *)(ht + row * CH_CP + q * 16);
        }
    }
    if (has3 && wave < 4) {
        const int f = wave, n3 = p.n3[br];
        floatx16 acc3;
        {
            floatx4 b3[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b3[q] = *(const floatx4 *)(p.b3[br] + 8 * q + 4 * fh);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = b3[r >> 2][r & 3];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const half8 a = *(const half8 *)(ht + (f * 32 + fm) * CH_CP + s * 32 + fh * 16);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], a, acc3, 0, 0, 0);
        }
        const int px = f * 32 + fm;
        float *o3 = p.out3[br] + (size_t)b * n3 * (CH_WO * CH_WO) + band * CH_PIX + px;
        if (px < CH_PIX) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (c < n3) o3[(size_t)c * (CH_WO * CH_WO)] = acc3[r];
            }
        }
    }
}

int launch_corr_head(const CorrHeadParams &p, void *stream) {
    if (!p.xs || !p.zk || !p.corr || !p.h0 || !p.w0_frag || !p.b0 || p.B < 1 || p.nb < 1 || p.nb > 3 || p.Cs < 256 * p.nb) return -1;
    hipLaunchKernelGGL(corr_head_kernel, dim3(CH_BR, p.nb, p.B), dim3(CH_NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
