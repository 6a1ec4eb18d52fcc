// conv_wreg.hip -- implicit-GEMM convolution with the WEIGHT operand loaded straight into registers.
//
// Same contraction as conv_igemm_kernel (C[m][n] = sum_k A[m][k] * W[n][k], BN folded, fused bias / residual / ReLU
// epilogue; experiments/siammask_sharp/resnet.py:64-103, models/rpn.py:45-60), different data path.  What bounds
// conv_igemm_kernel (measured in round 1): every consumer wave re-reads BOTH operands of its 64x64
// tile from LDS -- 4 ds_read_b128 (4 KB) per 4 MFMAs = 32 B per matrix-pipe cycle and wave, 128 B/clk for the four
// SIMDs of a CU, i.e. the whole LDS bandwidth, and the LDS-DMA of the next tile needs the same LDS ports (consumers
// alone reach 45 % of the MFMA peak with no loads at all; DMA + consumers 30-36 %).  Here only the ACTIVATION rows go
// through LDS (they are an im2col gather and are shared by all consumer waves of the workgroup); the weights are
// packed offline in MFMA-fragment order ("w_frag": one contiguous KB per (32 output channels, 16 k) fragment) and
// every consumer wave streams its own fragments global -> VGPR with fully coalesced 1 KB buffer loads, two K tiles
// ahead in a register ring (counted vmcnt by the compiler).  LDS reads per MFMA halve (FM = 2) or quarter per flop
// (FM = 4: 128 x 64 wave tiles), the LDS-DMA writes drop to the A rows, the weight stream never touches LDS.
//
//   workgroup  = 4 consumer waves (WN x WK) + 2 producer waves (A rows only, LDS-DMA, XOR swizzle on the source)
//   wave tile  = (32*FM) x 64;  workgroup tile = (32*FM) x (64*WN);  WK > 1 splits the k-steps of a K tile
//   K tile     = 128 B (64 halves), NSTAGE-deep A ring, ONE s_barrier per K tile (same protocol as conv_igemm_kernel)
// f16 only, NHWC epilogue only (the callers fall back to conv_igemm_kernel otherwise).
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

template <int A, int B> struct WMax { static constexpr int v = A > B ? A : B; };

template <int FM, int WN, int WK, int NSTAGE>
__global__ __launch_bounds__(384, (WMax<NSTAGE * 32 * FM * 128, 4 * 32 * FM * 68 * 4>::v <= 80 * 1024 ? 2 : 1))
void conv_wreg_kernel(const ConvBatch cb) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CONV_BATCH_MAX; ++i)
        if (i < cb.n && (int)blockIdx.x >= cb.start[i]) pi = i;
    const ConvParams &p = cb.p[pi];
    const int wg_first = cb.start[pi], wg_count = cb.start[pi + 1] - cb.start[pi];
    typedef _Float16 T;
    constexpr int NCW = 4, NPW = 2, NT = (NCW + NPW) * 64;
    static_assert(WN * WK == NCW, "four consumer waves");
    constexpr int BM = 32 * FM, BN = 64 * WN;
    constexpr int KT = 128, BK = 64, VE = 8;
    constexpr int RPR = NPW * 64 / 8;            // rows filled by one round of producer pieces (16)
    constexpr int RA = BM / RPR;                 // pieces per producer lane per K tile
    constexpr int NKS = 4 / WK;                  // k-steps (16 halves of K) per K tile per consumer
    constexpr int D = 2 * NKS;                   // weight fragments are loaded two K tiles ahead
    constexpr int AHEAD = NSTAGE - 1;
    constexpr int STAGE_BYTES = BM * KT;
    constexpr int LDE = 68;
    constexpr int EPI_BYTES = NCW * BM * LDE * 4;
    constexpr int LDS_BYTES = WMax<NSTAGE * STAGE_BYTES, EPI_BYTES>::v;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int cout_off = p.cout_off + g * p.g_cout_off;
    const float *bias = p.bias + g * p.g_wgt_off;

    const int tilesN = (p.Nst + BN - 1) / BN;
    int t = (int)blockIdx.x - wg_first;
    if (p.xcd_mode != 0) {                        // XCD-contiguous tm-major order (see conv_igemm_kernel)
        const int nblk = wg_count, q = nblk >> 3, r = nblk & 7;
        const int x = t & 7, j = t >> 3;
        t = x * q + (x < r ? x : r) + j;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.Kpad / BK;                   // K tiles (Kpad is a multiple of 128 elements: nk is even)

    floatx16 acc[FM][2];

    if (wave >= NCW) {
        // =========================== PRODUCER: gather the A rows, LDS-DMA =========================
        const int ptid = tid - NCW * 64, pw = wave - NCW;
        const int cin_off = p.cin_off + g * p.g_cin_off;
        const int lrow = ptid >> 3;
        const int slot = (ptid & 7) ^ ((lrow >> 1) & 7);
        RowInfo ri[RA];
        bool rvalid[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + lrow + RPR * i;
            rvalid[i] = m < p.M;
            ri[i] = row_info(p, rvalid[i] ? m : 0, p.pos);
        }
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
        constexpr long OOB = 0x7ffff000;
        const bool tap_uniform = p.ci_shift >= 0 && p.Ci >= BK;
        int cur_tap_s = -1, cur_c = 0;
        long a_off[RA];
        auto tap_offsets = [&](int tap) {
            const int kh_i = (tap * p.kw_magic) >> 16;
            const int kw_i = tap - kh_i * p.kw;
            const bool tap_ok = kh_i < p.kh;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ly = ri[i].ly0 + kh_i * p.dil, lx = ri[i].lx0 + kw_i * p.dil;
                bool ok = rvalid[i] & tap_ok & ((unsigned)ly < (unsigned)p.Hl) & ((unsigned)lx < (unsigned)p.Wl);
                int sy, sx;
                if (p.ups) {
                    sy = (ly * p.Hs) / p.Hl;
                    sx = (lx * p.Ws) / p.Wl;
                } else {
                    sy = ly + ri[i].oy_org;
                    sx = lx + ri[i].ox_org;
                }
                ok = ok & ((unsigned)sy < (unsigned)p.Hs) & ((unsigned)sx < (unsigned)p.Ws);
                const long off = (((long)(ri[i].b * p.Hs + sy) * p.Ws + sx) * p.Cs + cin_off) * (long)sizeof(T);
                a_off[i] = ok ? off : OOB;
            }
        };
        auto set_tile = [&](int kt) {
            if (tap_uniform) {
                const int k0 = kt * BK;
                const int tap = k0 >> p.ci_shift;
                cur_c = (k0 & (p.Ci - 1)) + slot * VE;
                if (tap != cur_tap_s) {
                    cur_tap_s = tap;
                    tap_offsets(tap);
                }
            } else {
                const int k = kt * BK + slot * VE;
                int tap;
                if (p.ci_shift >= 0) { tap = k >> p.ci_shift; cur_c = k & (p.Ci - 1); }
                else { tap = k / p.Ci; cur_c = k - tap * p.Ci; }
                tap_offsets(tap);
            }
        };
        auto issue_tile = [&](int buf) {
            unsigned char *sA = smem + buf * STAGE_BYTES + pw * 1024;
            const long cbyte = (long)cur_c * (long)sizeof(T);
#pragma unroll
            for (int j = 0; j < RA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t *)(sA + j * (RPR * KT)), 16,
                                                         (int)(a_off[j] + cbyte), 0, 0, 0);
        };
#pragma unroll
        for (int tt = 0; tt < AHEAD; ++tt)
            if (tt < nk) {
                set_tile(tt);
                issue_tile(tt);
            }
        int islot = AHEAD;
        if (islot == NSTAGE) islot = 0;
        for (int kt = 0; kt < nk; ++kt) {
            int younger = nk - 1 - kt;
            if (younger > AHEAD - 1) younger = AHEAD - 1;
            // at most `younger` whole tiles (RA pieces each) of this wave may still be in flight
            if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * RA) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + AHEAD < nk) {
                set_tile(kt + AHEAD);
                issue_tile(islot);
            }
            if (++islot == NSTAGE) islot = 0;
        }
    } else {
        // =========================== CONSUMER: A fragments from LDS, W fragments from global ======
        const int wk = wave % WK, wn = wave / WK;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int frow = lane & 31, fhalf = lane >> 5;
        const int fsw = (frow >> 1) & 7;
        const int a_row_off = frow * KT;
        // weight fragments: block (32 rows) nb, k-step k16 -> 1 KB at ((nb * KS16 + k16) * 64 + lane) * 16
        const int KS16 = p.Kpad >> 4;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt_frag, 0, p.w_bytes, 0x00020000);
        const int nb0 = (g * p.g_wgt_off + n0) / 32 + wn * 2;
        const int wv0 = (nb0 * KS16) * 1024 + lane * 16;        // voffset of fragment 0; fragment 1 is KS16 KB further
        const int wv1 = wv0 + KS16 * 1024;
        half8 fa[2][FM], fb[D][2];
        auto load_w = [&](int gstep, half8 (&b)[2]) {           // this wave's step gstep = (K tile, s): k16 = kt*4 + s*WK + wk
            const int kt = gstep / NKS, s = gstep - kt * NKS;
            const int so = (kt * 4 + s * WK + wk) * 1024;
            const uint4v x0 = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv0, so, 0);
            const uint4v x1 = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv1, so, 0);
            b[0] = __builtin_bit_cast(half8, x0);
            b[1] = __builtin_bit_cast(half8, x1);
        };
        auto read_a = [&](int buf, int s, half8 (&a)[FM]) {
            const unsigned char *sb = smem + buf * STAGE_BYTES + a_row_off;
            const int so = (((s * WK + wk) * 2 + fhalf) ^ fsw) << 4;
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = *(const half8 *)(sb + i * (32 * KT) + so);
        };
        auto mma_part = [&](const half8 (&a)[FM], const half8 (&b)[2], int q0, int q1) {
#pragma unroll
            for (int q = q0; q < q1; ++q)
                acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q >> 1], b[q & 1], acc[q >> 1][q & 1], 0, 0, 0);
        };
        const int S = nk * NKS;                                 // steps of this wave; a multiple of D
#pragma unroll
        for (int j = 0; j < D; ++j) load_w(j, fb[j]);
        __builtin_amdgcn_s_barrier();                           // barrier(0): A tile 0 is complete
        asm volatile("" ::: "memory");
        read_a(0, 0, fa[0]);
        int cur = 0;                                            // ring slot of the current A tile
        // one macro-iteration = D steps = two K tiles; REFILL: re-load the ring slot just consumed for step g + D
        auto body = [&](int g0, auto refill, auto last) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int s = j % NKS;
                if (s + 1 < NKS) {
                    read_a(cur, s + 1, fa[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(fa[j & 1], fb[j], 0, 2 * FM);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    // last k-step of a K tile: MFMA half | barrier(kt+1), first A fragments of the next tile | MFMA half
                    mma_part(fa[j & 1], fb[j], 0, FM);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(decltype(last)::value && j == D - 1)) {
                        int nxt = cur + 1;
                        if (nxt == NSTAGE) nxt = 0;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        cur = nxt;
                        read_a(cur, 0, fa[(j + 1) & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(fa[j & 1], fb[j], FM, 2 * FM);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (decltype(refill)::value) load_w(g0 + j + D, fb[j]);
            }
        };
        int g0 = 0;
        for (; g0 + D < S; g0 += D) body(g0, std::true_type{}, std::false_type{});
        body(g0, std::false_type{}, std::true_type{});
    }
    __syncthreads();

    // ---- epilogue: accumulators -> LDS -> (sum over the K-group) -> bias / residual / ReLU -> NHWC f16 ----------
    constexpr int EV = 8;
    constexpr int LPR = BN / EV;                         // threads per output row
    constexpr int RPP = NT / LPR;                        // rows per pass
    constexpr int NPASS = (BM + RPP - 1) / RPP;
    const int c4 = (tid % LPR) * EV, r0 = tid / LPR;
    const int n = n0 + c4;
    const bool ncol_ok = n < p.Nst;
    half8 rv[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int q = 0; q < EV; ++q) rv[ps][q] = (_Float16)0.f;
    if (p.res_mode != RES_NONE && ncol_ok) {
        const T *res = (const T *)p.res;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = ps * RPP + r0, m = m0 + row;
            if (row < BM && m < p.M) rv[ps] = *(const half8 *)(res + (size_t)m * p.res_Cs + p.res_coff + n);
        }
    }
    if (wave < NCW) {
        float *e = (float *)smem + wave * (BM * LDE);
        const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    e[row * LDE + j * 32 + frow] = acc[i][j][r];
                }
    }
    __syncthreads();
    if (!ncol_ok) return;
    const float *ecol = (const float *)smem + ((c4 >> 6) * WK) * (BM * LDE) + (c4 & 63);
    float bv[EV];
#pragma unroll
    for (int q = 0; q < EV; q += 4) {
        const floatx4 b4 = *(const floatx4 *)(bias + n + q);
        bv[q] = b4[0]; bv[q + 1] = b4[1]; bv[q + 2] = b4[2]; bv[q + 3] = b4[3];
    }
    T *out = (T *)p.out;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + r0, m = m0 + row;
        if (row < BM && m < p.M) {
            const float *er = ecol + row * LDE;
            half8 o;
#pragma unroll
            for (int q = 0; q < EV; q += 4) {
                floatx4 x = *(const floatx4 *)(er + q);
#pragma unroll
                for (int kq = 1; kq < WK; ++kq) x += *(const floatx4 *)(er + kq * (BM * LDE) + q);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v = x[u] + bv[q + u];
                    if (p.res_mode == RES_PRE_RELU) v += (float)rv[ps][q + u];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.res_mode == RES_POST_RELU) v += (float)rv[ps][q + u];
                    o[q + u] = (_Float16)v;
                }
            }
            *(half8 *)(out + (size_t)m * p.Cos + cout_off + n) = o;
        }
    }
}

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
    }
    for (int i = cb.n; i <= CONV_BATCH_MAX; ++i) cb.start[i] = total;
    dim3 grid(total, 1, groups);
    if (stages >= 4) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 4>), grid, dim3(384), 0, s, cb);
    else hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 3>), grid, dim3(384), 0, s, cb);
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
    if (bm == 64) {
        if (bn == 256) return launch_wreg_t<2, 4, 1>(cb, stages, s);
        if (bn == 128) return launch_wreg_t<2, 2, 2>(cb, stages, s);
        if (bn == 64) return launch_wreg_t<2, 1, 4>(cb, stages, s);
    } else if (bm == 128) {
        if (bn == 256) return launch_wreg_t<4, 4, 1>(cb, stages, s);
        if (bn == 128) return launch_wreg_t<4, 2, 2>(cb, stages, s);
        if (bn == 64) return launch_wreg_t<4, 1, 4>(cb, stages, s);
    }
    return 1;
}

}  // namespace smk
