// conv_wreg.hip -- implicit-GEMM convolution with the WEIGHT operand loaded straight into registers.
//
// Same contraction as conv_igemm_kernel (C[m][n] = sum_k A[m][k] * W[n][k], BN folded, fused bias / residual / ReLU
// epilogue; experiments/siammask_sharp/resnet.py:64-103, models/rpn.py:45-60), different data path.  In
// conv_igemm_kernel every consumer wave re-reads BOTH operands of its 64x64 tile from LDS (4 ds_read_b128 per 4 MFMAs)
// and both operands are staged by LDS-DMA.  Here only the ACTIVATION rows go through LDS (they are an im2col gather and
// are shared by all consumer waves of the workgroup); the weights are packed offline in MFMA-fragment order ("w_frag":
// one contiguous KB per (32 output channels, 16 k) fragment) and every consumer wave streams its own fragments
// global -> VGPR with fully coalesced 1 KB buffer loads, two K tiles ahead in a register ring (the compiler's counted
// vmcnt(13..15) waits, checked in the ISA).  LDS reads per MFMA halve (FM = 2) or quarter per flop (FM = 4: 128 x 64
// wave tiles), the LDS-DMA writes drop to the A rows, the weight stream never touches LDS.
// Measured with TWO producer waves (profiles/r02_wregbench_b8_b64.json): +5..15 % on the long-K N-wide layers, slower on
// short-K large-M layers -- LDS bandwidth (256 B/clk for ds_read_b128 on CDNA4) was NOT what bounded the LDS-staged kernel;
// DESIGN.md 3.1g.  With FOUR (a loader wave beside an MFMA-issuing wave is issue-bound): x1.06-1.64 per layer,
// faster than the best LDS-staged instantiation on almost every layer of the path at B = 1, 8 and 64.
//
//   workgroup  = 4 consumer waves (WN x WK) + NPW = 4 (or 2) producer waves (A rows only, LDS-DMA, XOR swizzle on the source)
//   wave tile  = (32*FM) x 64;  workgroup tile = (32*FM) x (64*WN);  WK > 1 splits the k-steps of a K tile
//   K tile     = 128 B (64 halves), NSTAGE-deep A ring, ONE s_barrier per K tile (same protocol as conv_igemm_kernel)
// f16 only, NHWC epilogue only (the callers fall back to conv_igemm_kernel otherwise).
//
// conv_seq_kernel (below) runs a whole SEQUENCE of such convolutions as ONE persistent launch: one workgroup per CU,
// the 32 workgroups of an XCD form a team that owns the images b = xcd, xcd + 8, ... and walks the layer list
// with team-local barriers; activations handed from layer to layer never leave the XCD's L2.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

template <int A, int B> struct WMax { static constexpr int v = A > B ? A : B; };

// per output row (b, oy, ox) of any parameter block with the ConvParams field names
template <class P> __device__ __forceinline__ RowInfo row_info_t(const P &p, int m) {
    RowInfo r;
    const int hw = p.Ho * p.Wo;
    r.b = m / hw;
    const int rem = m - r.b * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    r.ly0 = oy * p.stride - p.pad;
    r.lx0 = ox * p.stride_x - p.pad;
    r.oy_org = p.org_y;
    r.ox_org = p.org_x;
    if (p.pos) {
        r.oy_org += p.pos[2 * r.b + 0] * p.pos_mul + p.pos_add;
        r.ox_org += p.pos[2 * r.b + 1] * p.pos_mul + p.pos_add;
    }
    return r;
}

template <int FM, int NSTAGE, int NCW = 4> struct WregLds {
    static constexpr int v = WMax<NSTAGE * 32 * FM * 128, NCW * 32 * FM * 68 * 4>::v;
};

// ONE output tile rows [m0, min(m0 + BM, m_end)) x channels [n0, n0 + BN) of group g.  Called by all (4 + NPW) * 64 threads of
// the workgroup; starts and ends with the LDS free.  AUX = cache policy of the ACTIVATION loads (A rows, residual): 0 in
// the one-conv-per-launch kernel, sc1 (16: served by the L2, never by this CU's L1) in the persistent sequence kernel,
// where those bytes were written by another CU of the same XCD a moment ago.
//
// (Measured and removed, profiles/r02_wreg_pf_wave.txt: a seventh wave that touched the weight panel's lines 4-32 K tiles
// ahead of the consumers to warm the XCD's L2 -- no effect on any layer, so the K-tile time is not first-touch L2 latency.)
// (CLK = 1, measurement build of conv_seq_kernel only: thread 0 stamps the phases of the tile into tclk[0..6] -- entry, first
//  activation tile in LDS, K loop done, workgroup past the loop, accumulators handed over, stores issued, tile done.  CLK = 0
//  compiles to exactly the code without it.)
template <int FM, int WN, int WK, int NSTAGE, int AUX, int WT = 2, int NPW = 2, int CLK = 0, class P = ConvParams>
__device__ __forceinline__ void wreg_tile(const P &p, const int g, const int m0, const int m_end, const int n0,
                                          unsigned char *smem, unsigned long long *tclk = nullptr) {
    typedef _Float16 T;
    // NPW producer waves (2 or 4; smk_tune "npw"): tools/dma_patterns.hip measured that ONE loader wave beside MFMA waves
    // sustains a fixed ~8-14 GB/s of LDS-DMA whatever it has in flight, and that the rate of a CU grows with the number of
    // loader waves (2 -> 4 waves: x2) -- the activation stream of a 64-row tile is issue-bound on two producer waves.
    // Measured (profiles/r02_producer_waves_2_vs_4.txt): four producers x1.06-1.64 per layer, -7..8 % on the B=8 step, outputs
    // bit-identical.  Eight (64-row tiles, 12 waves, one workgroup per CU): +0-8 % on some layers, -15 % on layer1, step
    // unchanged -> removed (profiles/r02_producer_waves_4_vs_8.txt).
    constexpr int NCW = WN * WK, NT = (NCW + NPW) * 64;
    static_assert(NPW == 2 || NPW == 4, "two or four producer waves");
    // (eight consumers -- two MFMA-issuing waves per SIMD in one workgroup, 64x256 as 4x2 and 64x128 as 2x4 -- compile and
    // pass parity with this routine; measured 0-12 % slower than four on every layer, profiles/r02_wreg_ncw8.txt)
    static_assert(NCW == 4 && (WK == 1 || WK == 2 || WK == 4), "four consumer waves");
    constexpr int BM = 32 * FM, BN = 64 * WN;
    constexpr int KT = 128, BK = 64, VE = 8;
    constexpr int RPR = NPW * 64 / 8;            // rows filled by one round of producer pieces (16)
    constexpr int RA = BM / RPR;                 // pieces per producer lane per K tile
    constexpr int NKS = 4 / WK;                  // k-steps (16 halves of K) per K tile per consumer
    constexpr int D = WT * NKS;                  // weight fragments are loaded WT (= 2) K tiles ahead (four ahead + a 6-deep A ring
                                                 // measured 0-15 % slower on every layer: profiles/r02_wreg_deep_ring.txt)
    constexpr int AHEAD = NSTAGE - 1;
    constexpr int STAGE_BYTES = BM * KT;
    constexpr int LDE = 68;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto stamp = [&](int i) {
        if constexpr (CLK != 0) {
            if (tclk && tid == 0) tclk[i] = wall_clock64();
        }
    };
    stamp(0);
    const int cout_off = p.cout_off + g * p.g_cout_off;
    const float *bias = p.bias + g * p.g_wgt_off;
    const int nk = p.Kpad / BK;                   // K tiles (Kpad is a multiple of 128 elements: nk is even)

    floatx16 acc[FM][2];

    if (wave >= NCW) {
        // =========================== PRODUCER: gather the A rows, LDS-DMA =========================
        const int ptid = tid - NCW * 64, pw = wave - NCW;
        const int cin_off = p.cin_off + g * p.g_cin_off;
        const int lrow = ptid >> 3;
        // LDS-DMA writes lane-linear, so the XOR swizzle of the fragment reads goes on the SOURCE address (the eight lanes of
        // a row read its 128-byte line in permuted order).  p.a_stage = 1 (smk_tune "a_stage", A/B knob): the lanes read the
        // line in ascending order into registers and the swizzle is applied by a ds_write_b128 instead (same LDS image).
        const int wslot = (ptid & 7) ^ ((lrow >> 1) & 7);
        const int slot = p.a_stage ? (ptid & 7) : wslot;
        RowInfo ri[RA];
        bool rvalid[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + lrow + RPR * i;
            rvalid[i] = m < m_end;
            ri[i] = row_info_t(p, rvalid[i] ? m : m0);
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
                                                         (int)(a_off[j] + cbyte), 0, 0, AUX & 0xff);
        };
        if (p.a_stage) {
            // ---- register-staged variant: two K tiles in flight in VGPRs, written to the ring slot of tile kt right before
            // barrier(kt) (that slot held tile kt - NSTAGE, which nobody reads since barrier(kt - NSTAGE + 1)).  nk is even.
            uint4v rg[2][RA];
            auto load_regs = [&](uint4v (&r)[RA]) {
                const long cbyte = (long)cur_c * (long)sizeof(T);
#pragma unroll
                for (int j = 0; j < RA; ++j)
                    r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(a_off[j] + cbyte), 0, AUX & 0xff);
            };
            auto store_regs = [&](const uint4v (&r)[RA], int buf) {
                unsigned char *sA = smem + buf * STAGE_BYTES + lrow * KT + wslot * 16;
#pragma unroll
                for (int j = 0; j < RA; ++j) *(uint4v *)(sA + j * (RPR * KT)) = r[j];
            };
            set_tile(0);
            load_regs(rg[0]);
            set_tile(1);
            load_regs(rg[1]);
            int buf = 0;
            for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    store_regs(rg[h], buf);                 // the compiler's vmcnt wait covers exactly this tile's loads
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (kt + h + 2 < nk && !(AUX & 0x100)) {
                        set_tile(kt + h + 2);
                        load_regs(rg[h]);
                    }
                    if (++buf == NSTAGE) buf = 0;
                }
            }
        } else {
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
            if (kt + AHEAD < nk && !(AUX & 0x100)) {        // (0x100: measurement build without A refills)
                set_tile(kt + AHEAD);
                issue_tile(islot);
            }
            if (++islot == NSTAGE) islot = 0;
        }
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
            for (int q = q0; q < q1; ++q) {
                if (AUX & 0x400) {                          // (0x400: measurement build without the matrix pipe)
                    asm volatile("" ::"v"(a[q >> 1]), "v"(b[q & 1]));
                    continue;
                }
                acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q >> 1], b[q & 1], acc[q >> 1][q & 1], 0, 0, 0);
            }
        };
        const int S = nk * NKS;                                 // steps of this wave; a multiple of D
#pragma unroll
        for (int j = 0; j < D; ++j) load_w(j, fb[j]);
        __builtin_amdgcn_s_barrier();                           // barrier(0): A tile 0 is complete
        asm volatile("" ::: "memory");
        stamp(1);
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
                if (decltype(refill)::value && !(AUX & 0x200)) load_w(g0 + j + D, fb[j]);   // (0x200: no W refills)
            }
        };
        int g0 = 0;
        for (; g0 + D < S; g0 += D) body(g0, std::true_type{}, std::false_type{});
        body(g0, std::false_type{}, std::true_type{});
        stamp(2);
    }
    __syncthreads();
    stamp(3);

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
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = ps * RPP + r0, m = m0 + row;
            if (row < BM && m < m_end) {
                const uint4v x = __builtin_amdgcn_raw_buffer_load_b128(
                    rs_res, (int)(((size_t)m * p.res_Cs + p.res_coff + n) * sizeof(T)), 0, AUX & 0xff);
                rv[ps] = __builtin_bit_cast(half8, x);
            }
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
    stamp(4);
    if (ncol_ok) {
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
            if (row < BM && m < m_end) {
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
    stamp(5);
    __syncthreads();                                     // the LDS is free again (the next tile's producers may start)
    stamp(6);
}

// ---- one convolution (or a merged batch of independent ones) per launch ------------------------------------------
template <int FM, int WN, int WK, int NSTAGE, int NPW>
__global__ __launch_bounds__((WN * WK + NPW) * 64, ((NPW == 2 && WregLds<FM, NSTAGE, WN * WK>::v <= 80 * 1024) ? 2 : 1))
void conv_wreg_kernel(const ConvBatch cb) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CONV_BATCH_MAX; ++i)
        if (i < cb.n && (int)blockIdx.x >= cb.start[i]) pi = i;
    const ConvParams &p = cb.p[pi];
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
    wreg_tile<FM, WN, WK, NSTAGE, 0, 2, NPW>(p, (int)blockIdx.z, tm * BM, p.M, tn * BN, smem);
}

// measurement builds (smk_tune "ablate" = 1 no A refills, 2 no W refills, 4 no MFMA; results are wrong by construction)
template <int FM, int WN, int WK, int ABL>
__global__ __launch_bounds__(384, (WregLds<FM, 3>::v <= 80 * 1024 ? 2 : 1))
void conv_wreg_ablate_kernel(const ConvBatch cb) {
    const ConvParams &p = cb.p[0];
    constexpr int BM = 32 * FM, BN = 64 * WN;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WregLds<FM, 3>::v];
    const int tilesN = (p.Nst + BN - 1) / BN;
    const int t = (int)blockIdx.x;
    const int tm = t / tilesN, tn = t - tm * tilesN;
    wreg_tile<FM, WN, WK, 3, (ABL << 8)>(p, 0, tm * BM, p.M, tn * BN, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// conv_seq_kernel: a sequence of convolutions (a ResNet stage: Bottleneck after Bottleneck) as ONE persistent launch.
//
// Why: at B = 8 the step is ~50 dependent launches of 7-20 us.  Every launch boundary costs 1.5-2 us plus the
// write-back of what the predecessor left dirty (B / 6 TB/s), the grid fill / drain and each workgroup's cold start,
// and the next layer re-reads its input from the fabric because it was produced under other XCDs' L2s.  MI355X is
// eight XCDs with a private 4 MB L2 each -- and the workload is B independent images.  So: image b belongs to XCD
// b % 8 for the WHOLE sequence.  The 32 workgroups of an XCD (one per CU; a workgroup reads its XCD from
// HW_REG_XCC_ID and draws a ticket inside the team) share the tiles of their images layer by layer; between dependent layers they meet at a TEAM-LOCAL
// barrier: plain stores (they stay in the XCD's L2) -> s_waitcnt vmcnt(0) -> one L2-executed atomic per workgroup ->
// sc1 polls.  No agent-scope release / acquire, no L2 write-back, no L1 invalidate: the consumers read the handed-over
// activations with sc1 loads (L2-served), and a layer's 2 MB of activations are L2 hits for the next one.
// Weights are read-only (plain loads).  The kernel boundary at the end publishes the results to everybody else.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void team_barrier(unsigned *cnt, unsigned target, int *err) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // this wave's stores have reached the L2
    __syncthreads();                                                   // ... and every wave's of this workgroup
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // executes in the XCD's L2
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // sc1 load: L2-served
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 20000000ull) {                   // 0.2 s at 100 MHz: never hang the GPU
                atomicExch(err, 2);
                break;
            }
        }
    }
    __syncthreads();
}

template <int NPW, int CLK = 0>
__global__ __launch_bounds__((4 + NPW) * 64, 1) void conv_seq_kernel(const SeqArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WregLds<4, 3>::v];
    // team = the XCD this workgroup really runs on (HW_REG_XCC_ID; the dispatcher deals consecutive blocks round-robin
    // over the XCDs, starting wherever the previous launch stopped, so blockIdx says nothing); slot = arrival ticket
    // inside the team.  A one-block-per-CU launch puts gridDim/8 workgroups on every XCD (checked at smk_create).
    const int nslots = gridDim.x >> 3;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int team = (int)(xcc & 7);
    unsigned *cnt = a.bar + team * 32;                   // one 128-byte line per team: [0] barrier, [1] exits, [2] tickets
    int *slot_sh = (int *)smem;
    if (threadIdx.x == 0) *slot_sh = (int)__hip_atomic_fetch_add(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const int slot = *slot_sh;
    __syncthreads();
    if (slot >= nslots) {                                // more workgroups on this XCD than the census promised
        if (threadIdx.x == 0) atomicExch(a.err, 1);
        return;
    }
    unsigned nbar = 0;
    const bool clk = a.clk && team == 0 && slot == 0 && threadIdx.x == 0;
    if (clk) a.clk[0] = wall_clock64();
    for (int li = 0; li < a.n; ++li) {
        const SeqLayer &L = a.L[li];
        const int cfg = L.cfg;
        const int bn = (cfg == 0 || cfg == 3) ? 256 : ((cfg == 1 || cfg == 4) ? 128 : 64);
        const int bm = cfg >= 3 ? 128 : 64;
        const int tilesN = (L.Nst + bn - 1) / bn;
        const int hw = L.Ho * L.Wo;
        const int tiles = ((hw + bm - 1) / bm) * tilesN;
        for (int img = team; img < a.B; img += 8)
            for (int t = slot; t < tiles; t += nslots) {
                const int tm = t / tilesN, tn = t - tm * tilesN;
                const int m0 = img * hw + tm * bm, m_end = (img + 1) * hw;
                // (measurement build: the phases of this workgroup's FIRST tile of the layer, team 0 / slot 0)
                unsigned long long *tclk = nullptr;
                if constexpr (CLK != 0) tclk = (a.clk2 && team == 0 && slot == 0 && img == team && t == slot) ? a.clk2 + 8 * li : nullptr;
                if (cfg == 0) wreg_tile<2, 4, 1, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 256, smem, tclk);
                else if (cfg == 1) wreg_tile<2, 2, 2, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk);
                // 128-row tiles: weight fragments ONE K tile ahead (a k-step is 8 MFMAs here, so the cover in time is that of
                // two tiles at 64 rows; two ahead would need 234 + VGPRs and spill under this kernel's 256)
                else if (cfg == 3) wreg_tile<4, 4, 1, 3, 16, 1, NPW, CLK>(L, 0, m0, m_end, tn * 256, smem, tclk);
                else if (cfg == 4) wreg_tile<4, 2, 2, 3, 16, 1, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk);
                else wreg_tile<2, 1, 4, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 64, smem, tclk);
            }
        if (clk) a.clk[1 + 2 * li] = wall_clock64();
        if (L.sync && li + 1 < a.n) {
            ++nbar;
            team_barrier(cnt, nbar * (unsigned)nslots, a.err);
        }
        if (clk) a.clk[2 + 2 * li] = wall_clock64();
    }
    // the counters return to zero for the next launch: the LAST workgroup of the team to leave resets them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (prev == (unsigned)nslots - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// census: which XCD does block i run on?  (smk_create checks the i % 8 assumption once per context)
__global__ void xcc_census_kernel(int *out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf);
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
    if constexpr ((FM == 2 && WN == 2 && WK == 2) || (WN == 4 && WK == 1))      // measurement builds: three tile shapes only
    if (g_tune.ablate && cb.n == 1 && groups == 1) {
        switch (g_tune.ablate) {
        case 1: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 1>), grid, dim3(384), 0, s, cb); break;
        case 2: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 2>), grid, dim3(384), 0, s, cb); break;
        case 3: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 3>), grid, dim3(384), 0, s, cb); break;
        case 4: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 4>), grid, dim3(384), 0, s, cb); break;
        default: hipLaunchKernelGGL((conv_wreg_ablate_kernel<FM, WN, WK, 7>), grid, dim3(384), 0, s, cb); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    if (g_tune.npw == 4) {
        if (stages >= 4) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 4, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
        else hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 3, 4>), grid, dim3((WN * WK + 4) * 64), 0, s, cb);
    } else {
        if (stages >= 4) hipLaunchKernelGGL((conv_wreg_kernel<FM, WN, WK, 4, 2>), grid, dim3((WN * WK + 2) * 64), 0, s, cb);
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

int launch_conv_seq(const SeqArgs &a, int grid, void *stream) {
    if (a.n < 1 || a.n > SEQ_MAX || grid < 8 || (grid & 7) || !a.bar || !a.err) return -1;
    if (a.clk2 && g_tune.npw == 4)                        // SMK_SEQ_CLK=2: the build with the per-phase stamps (eager runs only)
        hipLaunchKernelGGL((conv_seq_kernel<4, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    else if (g_tune.npw == 4) hipLaunchKernelGGL(conv_seq_kernel<4>, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_seq_kernel<2>, dim3(grid), dim3(384), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int xcc_census(int grid, int *out_host) {
    int *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(int) * grid) != hipSuccess) return -4;
    hipLaunchKernelGGL(xcc_census_kernel, dim3(grid), dim3(384), 0, 0, d);
    hipError_t e = hipMemcpy(out_host, d, sizeof(int) * grid, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? 0 : -4;
}

}  // namespace smk
