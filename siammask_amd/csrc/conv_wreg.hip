// conv_wreg.hip -- implicit-GEMM convolution with the WEIGHT operand loaded straight into registers.
// (tile routine: wreg_tile.inc; persistent sequences: conv_seq.hip)
#include <hip/hip_runtime.h>
#include <type_traits>
#include "smk_kernels.h"

namespace smk {

#include "wreg_tile.inc"

// How far ahead the two operand streams run.  MODE 3 / 4: a 3- / 4-deep activation ring, weight fragments two K tiles ahead -- in K-STEPS
// (16 values of K: what one consumer wave multiplies between two refills) that is 8 + 8 for the N-wide tiles (WK = 1) but only 4 + 4 for WK = 2
// and 2 + 2 for WK = 4, because a K tile is 4 / WK steps of a wave.  The narrow tiles are the ones the path launches where M is small (B = 1:
// every layer; Refine's window convolutions; the split-operand contexts' layer3), i.e. where a launch is ONE latency chain per wave: two steps
// (~110 ns of MFMA) of cover against an L2 round trip.  MODE 8 (round 6) gives every shape the N-wide tiles' eight steps: ring 2 WK + 1 deep,
// weights 2 WK tiles ahead (WK = 1: identical to MODE 3).  MEASURED SLOWER wherever the narrow tiles run (profiles/r06r_wreg_deep_prefetch.txt:
// B = 1 step +9.6 %, B = 8 +2.5 %, f16x3 +2.9 %; the 64 x 64 launches +10..40 % each) -- like the six-deep ring of round 2
// (profiles/r02_wreg_deep_ring.txt): these launches are not waiting for operands that were requested too late.  `make MEASURE=1` builds only.
template <int WK, int MODE> struct WregDepth {
    static constexpr int NSTAGE = MODE == 8 ? 2 * WK + 1 : MODE;
    static constexpr int WT = MODE == 8 ? 2 * WK : 2;
};

// ---- one convolution (or a merged batch of independent ones) per launch ------------------------------------------
template <int FM, int WN, int WK, int MODE, int NPW>
__global__ __launch_bounds__((WN * WK + NPW) * 64, ((NPW == 2 && WregLds<FM, WregDepth<WK, MODE>::NSTAGE, WN * WK>::v <= 80 * 1024) ? 2 : 1))
void conv_wreg_kernel(const ConvBatch cb) {
    constexpr int NSTAGE = WregDepth<WK, MODE>::NSTAGE, WT = WregDepth<WK, MODE>::WT;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CONV_BATCH_MAX; ++i)
        if (i < cb.n && (int)blockIdx.x >= cb.start[i]) pi = i;
    const ConvParams &p = cb.p[pi];
    set_wave_prio(cb.p[0].wave_prio);
    const int wg_first = cb.start[pi], wg_count = cb.start[pi + 1] - cb.start[pi];
    constexpr int BM = 32 * FM, BN = 64 * WN;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WregLds<FM, NSTAGE, WN * WK>::v];
    const int tilesN = (p.Nst + BN - 1) / BN;
    int t = (int)blockIdx.x - wg_first;
    if (p.xcd_mode != 0) {                        // XCD-contiguous tm-major order (see conv_igemm_kernel)
        const int nblk = wg_count, q = nblk >> 3, r = nblk & 7;
        const int x = t & 7, j = t >> 3;
        t = x * q + (x < r ? x : r) + j;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    wreg_tile<FM, WN, WK, NSTAGE, 0, WT, NPW>(p, (int)blockIdx.z, tm * BM, p.M, tn * BN, smem);
}

#ifdef SMK_MEASURE
// measurement builds, only in a library built with `make MEASURE=1` (smk_tune "ablate": bit 1 no A refills, 2 no W refills, 4 no MFMA, 8 no K-loop barriers, 16 no A-fragment
// reads, 32 the first version's issue order (results right), 64 nothing removed; results are otherwise wrong by construction).  Four producer waves, like the production launches.
template <int FM, int WN, int WK, int ABL>
__global__ __launch_bounds__(512, 1)
void conv_wreg_ablate_kernel(const ConvBatch cb) {
    const ConvParams &p = cb.p[0];
    constexpr int BM = 32 * FM, BN = 64 * WN;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WregLds<FM, 3>::v];
    const int tilesN = (p.Nst + BN - 1) / BN;
    const int t = (int)blockIdx.x;
    const int tm = t / tilesN, tn = t - tm * tilesN;
    wreg_tile<FM, WN, WK, 3, (ABL << 8), 2, 4>(p, 0, tm * BM, p.M, tn * BN, smem);
}
#endif

// ---- dispatch -------------------------------------------------------------------------------------------------
template <int FM, int WN, int WK>
static int launch_wreg_t(ConvBatch &cb, int stages, hipStream_t s) {
    constexpr int BM = 32 * FM, BN = 64 * WN;
    int total = 0, groups = 1;
    for (int i = 0; i < cb.n; ++i) {
        ConvParams &p = cb.p[i];
        cb.start[i] = total;
        total += ((p.M + BM - 1) / BM) * ((p.Nst + BN - 1) / BN);
        if (p.groups > groups) groups = p.groups;
        p.ksplit = 1;
        if (p.x3_ct > 0 && p.x3_in > 0) {
            // split-operand pack in fused order (wreg_tile.inc x3ct): the producers gather the STORED channels [hi | lo] of a tap -- a plain fp16 gather over 2 C
            p.Ci = p.x3_in * X3_PLANES;
            p.ci_shift = -1;
            for (int sh = 0; sh < 16; ++sh)
                if ((1 << sh) == p.Ci) p.ci_shift = sh;
            p.x3_in = 0;
        }
    }
    for (int i = cb.n; i <= CONV_BATCH_MAX; ++i) cb.start[i] = total;
    dim3 grid(total, 1, groups);
#ifdef SMK_MEASURE
    if constexpr ((WN == 2 && WK == 2) || (WN == 4 && WK == 1))      // measurement builds: four tile shapes only
    if (g_tune.ablate && cb.n == 1 && groups == 1) {
        switch (g_tune.ablate) {
        case 1: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 1>), grid, dim3(512), 0, s, cb); break;
        case 2: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 2>), grid, dim3(512), 0, s, cb); break;
        case 3: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 3>), grid, dim3(512), 0, s, cb); break;
        case 4: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 4>), grid, dim3(512), 0, s, cb); break;
        case 7: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 7>), grid, dim3(512), 0, s, cb); break;
        case 8: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 8>), grid, dim3(512), 0, s, cb); break;
        case 16: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 16>), grid, dim3(512), 0, s, cb); break;
        case 32: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 32>), grid, dim3(512), 0, s, cb); break;
        case 64: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 64>), grid, dim3(512), 0, s, cb); break;
        case 11: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 11>), grid, dim3(512), 0, s, cb); break;
        case 27: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 27>), grid, dim3(512), 0, s, cb); break;
        default: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 23>), grid, dim3(512), 0, s, cb); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
#endif
    if (g_tune.npw == 4) {
#ifdef SMK_MEASURE
        if (stages == 8 && WK > 1) {
            hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, (WK > 1 ? 8 : 3), 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
            return hipGetLastError() == hipSuccess ? 0 : -4;
        }
        if (stages == 5 || stages == 6 || stages == 7) {          // deeper ACTIVATION rings only (weights two K tiles ahead as always)
            if (stages == 5) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 5, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
            else if (stages == 6) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 6, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
            else hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 7, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
            return hipGetLastError() == hipSuccess ? 0 : -4;
        }
#endif
        if (stages == 4) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 4, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
        else hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 3, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
    } else {
        if (stages == 4) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 4, 2>), grid, dim3((WN * WK + 2) * 64), 0, s, cb);
        else hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 3, 2>), grid, dim3((WN * WK + 2) * 64), 0, s, cb);
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

bool conv_wreg_eligible(const ConvParams &p, int dtype) {
    return dtype == DT_F16 && p.out_mode == OUT_NHWC && p.wgt_frag != nullptr && p.buf_lds && (p.Kpad % 128) == 0 &&
           (p.groups <= 1 || (p.g_wgt_off % 32) == 0);
}

// all problems of the batch: f16, NHWC epilogue, fragment-order weights present (conv_wreg_eligible)
int launch_conv_wreg_batch(ConvBatch &cb, int bm, int bn, int stages, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (cb.n < 1 || cb.n > CONV_BATCH_MAX) return -1;
    for (int i = 0; i < cb.n; ++i)
        if (!conv_wreg_eligible(cb.p[i], DT_F16) || (cb.p[i].groups > 1) != (cb.p[0].groups > 1)) return 1;
    if (bm == 32) {
        if (bn == 64) return launch_wreg_t<1, 1, 4>(cb, stages, s);      // (under-filled launches: twice the workgroups, half the activation rows each)
    } else if (bm == 64) {
        if (bn == 256) return launch_wreg_t<2, 4, 1>(cb, stages, s);
        if (bn == 128) return launch_wreg_t<2, 2, 2>(cb, stages, s);
        if (bn == 64) return launch_wreg_t<2, 1, 4>(cb, stages, s);
    } else if (bm == 96) {
        if (bn == 256) return launch_wreg_t<3, 4, 1>(cb, stages, s);
    } else if (bm == 128) {
        if (bn == 256) return launch_wreg_t<4, 4, 1>(cb, stages, s);
        if (bn == 128) return launch_wreg_t<4, 2, 2>(cb, stages, s);
        if (bn == 64) return launch_wreg_t<4, 1, 4>(cb, stages, s);
    }
    return 1;
}

}  // namespace smk
