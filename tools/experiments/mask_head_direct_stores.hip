// mask_head.hip -- the 63x63 mask head for LARGE batches (experiments/siammask_sharp/custom.py:89-96,185 -> models/rpn.py:56-61: the
// second convolution of MaskCorr's head, 1x1 256 -> 3969 + bias; experiments/siammask_base/custom.py the same), fp16 in, the
// [B, 3969, 25, 25] float32 NCHW tensor the reference returns out.
//
// Why its own kernel: at B = 64 the tensor is 633 MB -- 7.5 % of the step as an implicit-GEMM launch on 128 x 128 tiles at 2.6 TB/s
// (profiles/r05_b64_kernel_table.json), where plain streaming stores reach 4.9 TB/s on the same box (profiles/r04l_*).  The GEMM view
// has K = 256: a tile is two K tiles long, i.e. all prologue and epilogue, 10 016 of them, each re-staging its 64 KB of activations
// and weights through LDS and handing its accumulators through LDS again for the transposed store.  Here the ACTIVATIONS are
// stationary: a workgroup stages 128 pixels x 256 channels once (64 KB of LDS, two workgroups per CU), then walks the output
// channels in chunks of 128 -- each of the four waves one 32-channel block per chunk: weight fragments straight from the L2 into
// registers (the fragment-order pack conv_wreg_kernel uses; 2 MB, L2-resident), 64 MFMAs, and the accumulators go to memory from the
// registers: computed as W x A^T a register holds ONE channel's 32 consecutive pixels across the lanes, so a store instruction
// writes two full 128-byte runs of the NCHW tensor and nothing is transposed through LDS.  The stores of chunk i drain under the
// MFMAs of chunk i + 1.
//
// Same arithmetic as the implicit-GEMM path: fp16 operands, fp32 accumulation with v_mfma_f32_32x32x16_f16 in ascending k, bias added
// in fp32.  Used for batches beyond the chain_mask fusion (B > 16); the parity gates are the end-to-end ones (mask vs the oracle at
// B = 64) and tests/test_gpu_mask_head.py (against the implicit-GEMM launch on the same inputs).
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int MH_PX = 128;                 // pixels per workgroup
constexpr int MH_K = 256;                  // input channels (= K)
constexpr int MH_ROW = MH_K * 2;           // bytes per staged pixel row
constexpr int MH_CH = 128;                 // output channels per chunk (4 waves x 32)

__global__ __launch_bounds__(256, 2) void mask_head_kernel(const MaskHeadParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[MH_PX * MH_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (p.HW + MH_PX - 1) / MH_PX;
    int t = (int)blockIdx.x;
    const int ns = t % p.nsplit; t /= p.nsplit;
    const int pb = t % nblk, b = t / nblk;
    const int px0 = pb * MH_PX;

    // ---- the pixel block's activations -> LDS, 16-byte chunk c of row r at slot c ^ (r & 31): the 32 lanes of a fragment read (same
    // chunk, 32 different rows) hit 32 different slots
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.h0, 0, p.h0_bytes, 0x00020000);
        constexpr int NLD = MH_PX * (MH_ROW / 16) / 256;          // 16 loads per thread
        uint4v v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int g = i * 256 + tid, r = g >> 5, c = g & 31;
            const int px = px0 + r;
            const int ok = -(int)(px < p.HW);
            const int off = ((b * p.HW + px) * p.Cs + p.cin_off) * 2 + c * 16;
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (off & ok) | (0x7ffff000 & ~ok), 0, 0);      // beyond the image: zeros (range check)
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int g = i * 256 + tid, r = g >> 5, c = g & 31;
            *(uint4v *)(smem + r * MH_ROW + ((c ^ (r & 31)) << 4)) = v[i];
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_frag, 0, p.w_bytes, 0x00020000);
    constexpr int KS = MH_K / 16;                                  // 16 k-steps
    const int fm = lane & 31, fh = lane >> 5;
    const int chunks = (p.Npad / MH_CH + p.nsplit - 1) / p.nsplit;
    const int c_begin = ns * chunks;
    int c_end = c_begin + chunks;
    if (c_end > p.Npad / MH_CH) c_end = p.Npad / MH_CH;
    float *outb = p.out + (size_t)b * p.N * p.HW;
    half8 wf[KS];
    auto load_w = [&](int nb) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wf[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, ((nb * KS + s) * 64 + lane) * 16, 0, 0));
    };
    if (c_begin < c_end && (c_begin * 4 + wave) * 32 < p.N) load_w(c_begin * 4 + wave);
    for (int c = c_begin; c < c_end; ++c) {
        const int nb = c * 4 + wave;                               // this wave's 32-channel block
        if (nb * 32 >= p.N) break;                                 // (blocks of a wave ascend: nothing valid behind this one)
        floatx16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            half8 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 32 * j + fm;
                a[j] = *(const half8 *)(smem + row * MH_ROW + ((((2 * s + fh)) ^ (row & 31)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], a[j], acc[j], 0, 0, 0);
        }
        // the next chunk's weight fragments travel under this chunk's stores
        const int nbn = (c + 1) * 4 + wave;
        if (c + 1 < c_end && nbn * 32 < p.N) load_w(nbn);
        // ---- accumulators -> NCHW f32: register r of a lane = channel (r & 3) + 8 (r >> 2) + 4 fh of the block, the lane's pixel ----
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.bias[nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];        // (the pack pads the bias with zeros)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int px = px0 + 32 * j + fm;
            if (px < p.HW) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (ch < p.N) __builtin_nontemporal_store(acc[j][r] + bv[r], outb + (size_t)ch * p.HW + px);
                }
            }
        }
    }
}

int launch_mask_head(const MaskHeadParams &p, void *stream) {
    if (!p.h0 || !p.w_frag || !p.bias || !p.out || p.B < 1 || p.K != MH_K || p.Npad % MH_CH || p.N > p.Npad || p.nsplit < 1) return -1;
    const int nblk = (p.HW + MH_PX - 1) / MH_PX;
    hipLaunchKernelGGL(mask_head_kernel, dim3(p.B * nblk * p.nsplit), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
