// conv_igemm.hip -- implicit-GEMM convolution on the gfx950 matrix cores (MFMA).
//
// One kernel family serves every convolution on the SiamMask inference path
// (experiments/siammask_sharp/resnet.py:64-76,154 ; models/rpn.py:45-60 ;
//  experiments/siammask_sharp/custom.py:102-124,133-135,145-152):
//   C[m][n] = sum_k A[m][k] * W[n][k]        m = (b, oy, ox), n = cout, k = (kh, kw, cin)
// A is gathered on the fly from the NHWC activation tensor (zero padding, stride, dilation,
// crop/window origin per batch item, nearest-neighbour upsampling are all folded into the
// gather), W is the BN-folded weight matrix packed [Npad][Kpad].  The epilogue fuses
// bias (+ residual) (+ ReLU) and writes NHWC in the activation dtype, or NCHW fp32 for the
// tensors handed back to the caller.
//
// Structure (details at the kernel): workgroup = 4 or 8 CONSUMER waves (each owns a 64x64
// accumulator tile of 2x2 MFMA 32x32 fragments; f16: v_mfma_f32_32x32x16_f16, f32:
// v_mfma_f32_32x32x2_f32, an exact fp32 fma chain) + 4 PRODUCER waves that gather the A rows and the
// weight rows straight into an LDS ring with LDS-DMA (global_load_lds_dwordx4, 16 bytes per lane,
// XOR-swizzled on the source side so that the ds_read_b128 fragment reads are bank-conflict free).
// One s_barrier per K tile hands a tile from the producers to the consumers.  The accumulators go
// back through LDS once so that global stores (and residual loads) are 16-byte row-contiguous.
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <typename T> struct Traits;
template <> struct Traits<float> {
    static constexpr int VE = 4;                 // elements per 16-byte vector
    typedef floatx4 frag_t;
    typedef floatx4 out4_t;
    static __device__ inline void mma(floatx16 &acc, const frag_t &a, const frag_t &b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
    }
    static __device__ inline floatx4 load4(const float *p) { return *(const floatx4 *)p; }
    static __device__ inline void store4(float *p, floatx4 v) { *(floatx4 *)p = v; }
    // 16 bytes of output channels per thread (EV = 4)
    static constexpr int EV = 4;
    static __device__ inline void loadv(const float *p, float (&v)[4]) {
        floatx4 x = *(const floatx4 *)p;
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    }
    static __device__ inline void storev(float *p, const float (&v)[4]) {
        floatx4 x = {v[0], v[1], v[2], v[3]};
        *(floatx4 *)p = x;
    }
};
template <> struct Traits<_Float16> {
    static constexpr int VE = 8;
    typedef half8 frag_t;
    typedef half4 out4_t;
    static __device__ inline void mma(floatx16 &acc, const frag_t &a, const frag_t &b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    static __device__ inline floatx4 load4(const _Float16 *p) {
        half4 h = *(const half4 *)p;
        floatx4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        return v;
    }
    static __device__ inline void store4(_Float16 *p, floatx4 v) {
        half4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *(half4 *)p = h;
    }
    // 16 bytes of output channels per thread (EV = 8)
    static constexpr int EV = 8;
    static __device__ inline void loadv(const _Float16 *p, float (&v)[8]) {
        half8 h = *(const half8 *)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
    }
    static __device__ inline void storev(_Float16 *p, const float (&v)[8]) {
        half8 h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
        *(half8 *)p = h;
    }
};

template <int A, int B> struct CMax { static constexpr int v = A > B ? A : B; };

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// 16-byte direct-to-LDS load (LDS-DMA): lane l's 16 bytes land at lds_base + 16*l.
// lds_base must be wave-uniform (it travels in M0); the global source is per lane.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned char *lds_base) {
    __builtin_amdgcn_global_load_lds((gbl_void_t *)gsrc, (lds_void_t *)lds_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n_tiles * NP of this wave's LDS-DMA pieces are still in flight
template <int NP> __device__ __forceinline__ void wait_tiles(int n_tiles) {
    if (n_tiles <= 0) wait_vmcnt<0>();
    else if (n_tiles == 1) wait_vmcnt<NP>();
    else wait_vmcnt<2 * NP>();
}

// XOR swizzle of the 16-byte slot inside an LDS row (SPR slots per row): makes the ds_read_b128
// fragment reads (32 consecutive rows, same logical slot) bank-conflict free.
//   SPR = 8  (128-byte rows, two rows per 256-byte bank line): (row>>1)&7
//   SPR = 16 (256-byte rows, one row per bank line):            row&15
template <int SPR> __device__ __forceinline__ int swz(int row) {
    return SPR == 8 ? ((row >> 1) & 7) : (row & 15);
}

// ---------------------------------------------------------------------------------------------
// conv_igemm_kernel<T, WM, WN, WK, KT, OUT_MODE, NSTAGE>
//   (NCW + 4) waves = NCW CONSUMER waves (4 or 8) + 4 PRODUCER waves (wave specialisation).
//   Measured on the previous all-waves-do-everything kernel (profiles/r01_v3_ablation_128x128.json):
//   LDS-DMA issue, LDS fragment reads and MFMAs serialise inside an in-order wave
//   (B=64 l3.0.ds: DMA alone 452 us, MFMA + reads alone 511 us, together 754 us).  Here the
//   producers only gather addresses and issue the LDS-DMA, the consumers only read fragments and
//   issue MFMAs, so a producer stalled on the memory pipe never blocks the matrix pipe.
//
//   Consumers are arranged WM x WN x WK (WM*WN*WK == 4 or 8); EVERY consumer owns a 64x64 accumulator
//   tile (2x2 MFMA 32x32 fragments), so the LDS->register traffic per MFMA is the same for all
//   workgroup shapes (4 ds_read_b128 per 4 MFMAs).  Workgroup tile = 64*WM x 64*WN; when WK > 1
//   the consumers of a K-group split the k-steps of every K tile between them and the partial
//   sums are added in the epilogue (through LDS), so small tiles stay LDS-efficient.
//   K tile = KT bytes of K per row (128 or 256), staged into an NSTAGE-deep LDS ring.
//
//   Hand-over protocol, ONE s_barrier per K tile for all eight waves.  barrier(kt) means
//   "tile kt is complete in LDS and nobody reads tile kt-1 any more":
//     producer: wait vmcnt(own pieces of tile kt) ; barrier(kt) ; issue tile kt+NSTAGE-1 into the
//               slot tile kt-1 occupied
//     consumer: barrier(kt) ; read fragments of tile kt / MFMA (its reads of tile kt are all
//               consumed by MFMAs before it reaches barrier(kt+1))
// ---------------------------------------------------------------------------------------------
template <int STAGE_BYTES, int NSTAGE, int EPI, int NWAVES> struct WgPerCu {
    static constexpr int lds = CMax<STAGE_BYTES * NSTAGE, EPI>::v;
    static constexpr int wgs = (160 * 1024 / lds) >= 2 ? 2 : 1;   // workgroups per CU the LDS admits (cap 2)
    static constexpr int waves_per_simd = wgs * NWAVES / 4;
};
#define SMK_NCW (WM * WN * WK)
#define SMK_EPI (SMK_NCW * 64 * 68 * 4)

template <int WM, int WN, int WK, int KT, int NSTAGE> struct IgemmLds {
    static constexpr int v = CMax<NSTAGE * 64 * (WM + WN) * KT, SMK_EPI>::v;
};

// the body of conv_igemm_kernel for (virtual) workgroup bx of group bz, run by the (NCW + 4) * 64 threads whose index is
// threadIdx.x - tid0, in `smem` (IgemmLds<...>::v bytes): a plain kernel calls it with (blockIdx.x, blockIdx.z, 0); the
// horizontally fused tail kernel (chain_mask_kernel) runs two of them side by side in one 1024-thread workgroup
template <typename T, int WM, int WN, int WK, int KT, int OUT_MODE, int NSTAGE>
__device__ __forceinline__ void conv_igemm_body(const ConvBatch &cb, const int bx, const int bz, const int tid0,
                                                unsigned char *smem) {
    // several independent convolutions can share one launch (same instantiation): workgroups
    // [start[i], start[i+1]) belong to problem i
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CONV_BATCH_MAX; ++i)
        if (i < cb.n && bx >= cb.start[i]) pi = i;
    const ConvParams &p = cb.p[pi];
    const int wg_first = cb.start[pi], wg_count = cb.start[pi + 1] - cb.start[pi];
    constexpr int NCW = SMK_NCW;               // consumer waves (4 or 8); 4 producer waves follow
    constexpr int NT = (NCW + 4) * 64;
    static_assert(NCW == 4 || NCW == 8, "four or eight consumer waves per workgroup");
    typedef Traits<T> TR;
    typedef typename TR::frag_t frag_t;
    constexpr int VE = TR::VE;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int BK = KT / (int)sizeof(T);
    constexpr int SPR = KT / 16;               // 16-byte slots per LDS row
    constexpr int RPR = 256 / SPR;             // rows filled by one round of 256 LDS-DMA pieces
    constexpr int RA = BM / RPR, RB = BN / RPR, NP = RA + RB;   // pieces per producer thread per K tile
    constexpr int NKS = (KT / 32) / WK;        // k-steps (32 bytes of K) per K tile per consumer
    static_assert(NKS >= 2 && NKS % 2 == 0, "register double buffer needs an even step count");
    constexpr int AHEAD = NSTAGE - 1;          // K tiles in flight
    static_assert(AHEAD >= 1 && AHEAD <= 3, "ring depth 2..4");
    constexpr int STAGE_BYTES = (BM + BN) * KT;
    constexpr int LDE = 68;                    // NHWC: row-major [64 rows][68]; NCHW: column-major [64 cols][68]
    constexpr int EPI_BYTES = NCW * 64 * LDE * 4;               // one 64x64 f32 accumulator tile per consumer
    static_assert(CMax<NSTAGE * STAGE_BYTES, EPI_BYTES>::v == IgemmLds<WM, WN, WK, KT, NSTAGE>::v, "LDS size");

    const int tid = (int)threadIdx.x - tid0, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = bz;
    const int cout_off = p.cout_off + g * p.g_cout_off;
    const float *bias = p.bias + g * p.g_wgt_off;

    // XCD-aware tile order (p.xcd_mode): workgroup b runs on XCD b % 8 (observed dispatch
    // order, used for speed only).  Mode 1 hands every XCD a contiguous range of the tm-major
    // tile sequence, so the workgroups sharing one activation row panel (same tm, all tn) run
    // on ONE XCD and that panel is fetched into one L2 only.
    const int tilesN = (p.Nst + BN - 1) / BN;
    int t = bx - wg_first;
    if (p.xcd_mode != 0) {
        const int nblk = wg_count, q = nblk >> 3, r = nblk & 7;
        const int x = t & 7, j = t >> 3;
        t = x * q + (x < r ? x : r) + j;
    }
    // split-K (p.ksplit > 1, NHWC epilogue only): `ksplit` consecutive workgroups share an output tile, each
    // runs 1/ksplit of the K loop; the one that finishes last adds the partial tiles up (see the epilogue).
    // Consecutive t -> same XCD under xcd_mode 1, so the partial tiles meet in one L2.
    const int ksplit = (OUT_MODE == OUT_NHWC && p.ksplit > 1) ? p.ksplit : 1;
    int ks = 0;
    if (ksplit > 1) {
        ks = t % ksplit;
        t = t / ksplit;
    }
    int tm, tn;
    if (p.xcd_mode == 2) {            // tn-major: each XCD owns a range of weight panels
        const int tilesM = (p.M + BM - 1) / BM;
        tn = t / tilesM; tm = t - tn * tilesM;
    } else {
        tm = t / tilesN; tn = t - tm * tilesN;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = ((p.K + BK - 1) / BK) / ksplit;     // K tiles of THIS workgroup (the host makes it divide)
    const int kt0 = ks * nk;                           // its first K tile

    floatx16 acc[2][2];                // consumers only

    if (wave >= NCW) {
        // =========================== PRODUCER: gather + LDS-DMA ===============================
        const int ptid = tid - NCW * 64, pw = wave - NCW;
        if (p.prio == -1) __builtin_amdgcn_s_setprio(1);    // (measurement: producers prioritised instead)
        const int cin_off = p.cin_off + g * p.g_cin_off;
        const T *wgt = (const T *)p.wgt + (size_t)g * p.g_wgt_off * p.Kpad;
        // LDS-DMA writes lane-linear: thread ptid fills (row ptid/SPR [+RPR*i], physical slot
        // ptid%SPR).  The XOR swizzle therefore goes on the SOURCE: this thread fetches the logical
        // 16-byte slot  phys ^ swz(row)  of its row, and the fragment reads apply the same XOR.
        const int lrow = ptid / SPR;
        const int slot = (ptid % SPR) ^ swz<SPR>(lrow);     // swz(lrow + RPR*i) == swz(lrow)
        RowInfo ri[RA];
        bool rvalid[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int m = m0 + lrow + RPR * i;
            rvalid[i] = m < p.M;
            ri[i] = row_info(p, rvalid[i] ? m : 0, p.pos);
        }
        const char *in = (const char *)p.in;
        const long zero_off = (const char *)p.zero - in;   // 16 KB of zeros: source of all padding
        const char *wsrc[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            wsrc[i] = (const char *)(wgt + (size_t)(n0 + lrow + RPR * i) * p.Kpad + slot * VE);
        // buffer-resource form of the same loads (p.buf_lds): 32-bit offsets against an SRD whose
        // hardware range check returns zeros for every out-of-range lane, so padding needs no
        // zero page and no memory traffic at all
        const bool use_buf = p.buf_lds != 0;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, p.w_bytes, 0x00020000);
        const long w_base = (const char *)wgt - (const char *)p.wgt;
        constexpr long OOB = 0x7ffff000;                   // >= num_records (tensors are < 2 GB in this mode)

        // This thread always fetches the same 16-byte slot of every K tile, i.e. K index
        // kt*BK + slot*VE.  Its (tap, channel) position is decoded per tile with a shift (Ci is a
        // power of two on this path; p.ci_shift < 0 selects the division fallback).  When a K tile
        // never straddles a tap (Ci >= BK) the tap is wave-uniform and the per-row source offsets
        // are recomputed only when it changes (scalar branch); otherwise they are recomputed per
        // tile.  Padding (conv zero padding, rows >= M, K tail) reads the zero page, so the loads
        // are branch-free.
        // The geometry the K loop needs, as OPAQUE scalar copies (see wreg_tile.inc: without this the compiler re-loads the
        // fields from the kernel-argument segment wherever the tap changes inside the loop, and waits for them)
        auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        int g_kwm = sgpr(p.kw_magic), g_kw = sgpr(p.kw), g_kh = sgpr(p.kh), g_dil = sgpr(p.dil), g_Hl = sgpr(p.Hl), g_Wl = sgpr(p.Wl),
            g_Hs = sgpr(p.Hs), g_Ws = sgpr(p.Ws), g_Cs = sgpr(p.Cs), g_cish = sgpr(p.ci_shift), g_Ci = sgpr(p.Ci), g_ups = sgpr(p.ups);
        const int g_x3in = sgpr(p.x3_in);                  // split tensors: operand channels [hi | hi | lo] of a tap -> stored planes [hi | lo]
        asm volatile("" : "+s"(g_kwm), "+s"(g_kw), "+s"(g_kh), "+s"(g_dil), "+s"(g_Hl), "+s"(g_Wl));
        asm volatile("" : "+s"(g_Hs), "+s"(g_Ws), "+s"(g_Cs), "+s"(g_cish), "+s"(g_Ci), "+s"(g_ups));
        // (round 6: a K tile never straddles a tap whenever Ci is a MULTIPLE of the tile, power of two or not -- the split-operand
        //  packs have Ci = 3 x 2^s; the tap is then one division per tile on wave-uniform values instead of one per lane and row)
        const bool tap_uniform = g_Ci >= BK && (g_cish >= 0 || (g_Ci % BK) == 0);
        int cur_tap_s = -1;                                // wave-uniform tap of the last decode
        int cur_c = 0;
        long a_off[RA];                                    // byte offsets relative to `in`
        auto tap_offsets = [&](int tap) {
            const int kh_i = (tap * g_kwm) >> 16;
            const int kw_i = tap - kh_i * g_kw;
            const bool tap_ok = kh_i < g_kh;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ly = ri[i].ly0 + kh_i * g_dil, lx = ri[i].lx0 + kw_i * g_dil;
                bool ok = rvalid[i] & tap_ok & ((unsigned)ly < (unsigned)g_Hl) & ((unsigned)lx < (unsigned)g_Wl);
                int sy, sx;
                if (g_ups) {                               // nearest upsampling (uniform branch)
                    sy = (ly * g_Hs) / g_Hl;
                    sx = (lx * g_Ws) / g_Wl;
                } else {
                    sy = ly + ri[i].oy_org;
                    sx = lx + ri[i].ox_org;
                }
                ok = ok & ((unsigned)sy < (unsigned)g_Hs) & ((unsigned)sx < (unsigned)g_Ws);
                const long off = (((long)(ri[i].b * g_Hs + sy) * g_Ws + sx) * g_Cs + cin_off) * (long)sizeof(T);
                a_off[i] = ok ? off : (use_buf ? OOB : zero_off);
            }
        };
        auto set_tile = [&](int kt) {
            if (tap_uniform) {
                const int k0 = kt * BK;
                const int tap = g_cish >= 0 ? k0 >> g_cish : k0 / g_Ci;
                cur_c = (g_cish >= 0 ? (k0 & (g_Ci - 1)) : k0 - tap * g_Ci) + slot * VE;
                if (tap != cur_tap_s) {
                    cur_tap_s = tap;
                    tap_offsets(tap);
                }
            } else {
                const int k = kt * BK + slot * VE;
                int tap;
                if (g_cish >= 0) { tap = k >> g_cish; cur_c = k & (g_Ci - 1); }
                else { tap = k / g_Ci; cur_c = k - tap * g_Ci; }
                tap_offsets(tap);
            }
            if (g_x3in > 0 && cur_c >= g_x3in) cur_c -= g_x3in;
        };
        // all NP pieces of K tile kt into ring slot `buf`
        auto issue_tile = [&](int kt, int buf) {
            unsigned char *sA = smem + buf * STAGE_BYTES + pw * 1024;
            const long cb = (long)cur_c * (long)sizeof(T), kb = (long)kt * KT;
            if (use_buf) {
#pragma unroll
                for (int j = 0; j < RA; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t *)(sA + j * 4096), 16,
                                                             (int)(a_off[j] + cb), 0, 0, 0);
#pragma unroll
                for (int j = 0; j < RB; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t *)(sA + BM * KT + j * 4096), 16,
                                                             (int)(wsrc[j] - (const char *)wgt + w_base), (int)kb, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < RA; ++j) glds16(in + a_off[j] + cb, sA + j * 4096);
#pragma unroll
                for (int j = 0; j < RB; ++j) glds16(wsrc[j] + kb, sA + BM * KT + j * 4096);
            }
        };

#pragma unroll
        for (int tt = 0; tt < AHEAD; ++tt)
            if (tt < nk) {
                set_tile(kt0 + tt);
                issue_tile(kt0 + tt, tt);
            }
        int islot = AHEAD;                                 // slot of tile kt + AHEAD
        if (islot == NSTAGE) islot = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt: my pieces have landed; tiles kt+1 .. min(kt+AHEAD-1, nk-1) may still be in flight
            int younger = nk - 1 - kt;
            if (younger > AHEAD - 1) younger = AHEAD - 1;
            wait_tiles<NP>(younger);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + AHEAD < nk) {
                set_tile(kt0 + kt + AHEAD);
                issue_tile(kt0 + kt + AHEAD, islot);
            }
            if (++islot == NSTAGE) islot = 0;
        }
    } else {
        // =========================== CONSUMER: LDS fragments + MFMA ===========================
        const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int frow = lane & 31, fhalf = lane >> 5;
        const int fsw = swz<SPR>(frow);                    // swz(64*w + 32*i + frow) == swz(frow)
        const int a_row_off = (wm * 64 + frow) * KT;
        const int b_row_off = BM * KT + (wn * 64 + frow) * KT;
        frag_t fa[2][2], fb[2][2];                         // [step parity][fragment]
        auto read_frags = [&](int buf, int s, frag_t (&a)[2], frag_t (&b)[2]) {
            const unsigned char *sb = smem + buf * STAGE_BYTES;
            const int so = (((s * WK + wk) * 2 + fhalf) ^ fsw) << 4;
            a[0] = *(const frag_t *)(sb + a_row_off + so);
            a[1] = *(const frag_t *)(sb + a_row_off + 32 * KT + so);
            b[0] = *(const frag_t *)(sb + b_row_off + so);
            b[1] = *(const frag_t *)(sb + b_row_off + 32 * KT + so);
        };
        // the four MFMAs of a k-step, issued as [q0, q1) so that other work can be placed between them
        auto mma_part = [&](int par, int q0, int q1) {
#pragma unroll
            for (int q = q0; q < q1; ++q) TR::mma(acc[q >> 1][q & 1], fa[par][q >> 1], fb[par][q & 1]);
        };

        // matrix-pipe waves outrank the memory-issuing producers of the same SIMD (static priority)
        if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
        __builtin_amdgcn_s_barrier();                      // barrier(0): tile 0 is complete
        asm volatile("" ::: "memory");
        read_frags(0, 0, fa[0], fb[0]);
        // Issue order inside a k-step (pinned with sched_barrier):
        //   ds_read fragments of step s+1          (register double buffer)
        //   MFMA 0..3 of step s                    <- hipcc waits lgkmcnt(4): only for the fragments
        //                                             read one step ago; 128 cycles of matrix pipe
        //                                             cover the new reads
        // The last k-step of a tile carries the hand-over: MFMA 0,1 | barrier(kt+1), first
        // fragments of tile kt+1 | MFMA 2,3.
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            int nxt = cur + 1;
            if (nxt == NSTAGE) nxt = 0;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s == 0) {
                    // (hipcc waits lgkmcnt(0) at the loop head: keep the new reads behind MFMA 0)
                    mma_part(0, 0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(cur, 1, fa[1], fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(0, 1, 4);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (s + 1 < NKS) {
                    read_frags(cur, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(s & 1, 0, 4);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    mma_part(s & 1, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 1 < nk) {
                        // all my LDS reads of tile kt are complete (the MFMAs above consumed them)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();          // barrier(kt+1)
                        asm volatile("" ::: "memory");
                        read_frags(nxt, 0, fa[0], fb[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(s & 1, 2, 4);
                }
            }
            cur = nxt;
        }
    }
    __syncthreads();

    // ---- epilogue: the accumulator tiles -> LDS -> (sum over the K-group) -> fused
    //      bias/res/relu -> global, by all waves.  The residual rows are fetched first, so that
    //      their latency hides behind the accumulator hand-over through LDS. ----------------------
    constexpr int EV = TR::EV;                           // NHWC: output channels per thread (16 bytes)
    constexpr int LPR = BN / EV;                         // threads per output row
    constexpr int RPP = NT / LPR;                        // rows per pass
    constexpr int NPASS = (BM + RPP - 1) / RPP;
    const int c4 = (tid % LPR) * EV, r0 = tid / LPR;
    float rv[NPASS][EV];
    if (OUT_MODE == OUT_NHWC) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int q = 0; q < EV; ++q) rv[ps][q] = 0.f;
        if (p.res_mode != RES_NONE && n0 + c4 < p.Nst) {
            const T *res = (const T *)p.res;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = ps * RPP + r0, m = m0 + row;
                if (row < BM && m < p.M) {
                    TR::loadv(res + (size_t)m * p.res_Cs + p.res_coff + n0 + c4, rv[ps]);
                    if constexpr (sizeof(T) == 2) {
                        if (p.x3_res > 0) {                    // split residual (DT_F16X3): value = hi + lo
                            float lo[EV];
                            TR::loadv(res + (size_t)m * p.res_Cs + p.res_coff + p.x3_res + n0 + c4, lo);
#pragma unroll
                            for (int q = 0; q < EV; ++q) rv[ps][q] += lo[q];
                        }
                    }
                }
            }
        }
    }
    if (wave < NCW) {
        float *e = (float *)smem + wave * (64 * LDE);
        const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    int col = j * 32 + frow;
                    if (OUT_MODE == OUT_NHWC) {
                        e[row * LDE + col] = acc[i][j][r];
                    } else if ((r & 3) == 0) {
                        // column-major for the NCHW epilogue: r&3 = 0..3 are four consecutive rows
                        floatx4 v = {acc[i][j][r], acc[i][j][r + 1], acc[i][j][r + 2], acc[i][j][r + 3]};
                        *(floatx4 *)(e + col * LDE + row) = v;
                    }
                }
    }
    __syncthreads();
    const float *ebase = (const float *)smem;
    // region of consumer (wm, wn, wk): ((wm*WN + wn)*WK + wk) * 64*LDE floats
    if (OUT_MODE == OUT_NHWC) {
        const int n = n0 + c4;
        const bool ncol_ok = n < p.Nst;
        // K-group sums of this thread's NPASS x EV outputs
        float vv[NPASS][EV];
        {
            const float *ecol = ebase + ((c4 >> 6) * WK) * (64 * LDE) + (c4 & 63);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = ps * RPP + r0;
                const float *er = ecol + ((row >> 6) * WN * WK) * (64 * LDE) + (row & 63) * LDE;
#pragma unroll
                for (int q = 0; q < EV; q += 4) {
                    floatx4 x = {0.f, 0.f, 0.f, 0.f};
                    if (row < BM) {
                        x = *(const floatx4 *)(er + q);
#pragma unroll
                        for (int kq = 1; kq < WK; ++kq) x += *(const floatx4 *)(er + kq * (64 * LDE) + q);
                    }
                    vv[ps][q] = x[0]; vv[ps][q + 1] = x[1]; vv[ps][q + 2] = x[2]; vv[ps][q + 3] = x[3];
                }
            }
        }
        if (ksplit > 1) {
            // partial tiles in thread-linear order [tile][ks][pass][thread][EV] f32: every thread later reads
            // back exactly the addresses its counterparts wrote (coalesced both ways).  The workgroup that
            // arrives last at the tile's counter sums ALL partials in the fixed order ks = 0..ksplit-1
            // (its own included), so the result does not depend on the arrival order.
            const size_t tile_id = (size_t)tm * tilesN + tn;
            // No fences: an agent-scope release / acquire would write back and invalidate the whole L2 of the
            // XCD per workgroup (measured: +35 % on the B=8 step).  The partial tiles travel with sc1 (device
            // coherent: write-through / L2-bypassing) buffer accesses instead, `s_waitcnt vmcnt(0)` makes the
            // stores complete before the arrival counter is bumped, and the counter is a device-scope atomic.
            constexpr int PART = NPASS * NT * EV;                        // f32 per (tile, part)
            typedef unsigned uint4v __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(p.ks_part + tile_id * ksplit * PART), 0, (int)(ksplit * PART * sizeof(float)), 0x00020000);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                for (int q = 0; q < EV; q += 4) {
                    const uint4v d = {__float_as_uint(vv[ps][q]), __float_as_uint(vv[ps][q + 1]),
                                      __float_as_uint(vv[ps][q + 2]), __float_as_uint(vv[ps][q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs_part, (int)((((ks * NPASS + ps) * NT + tid) * EV + q) * sizeof(float)), 0, 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my part of the partial tile is in memory
            __syncthreads();                                   // ... and everybody's; LDS is free from here on
            int *flag = (int *)smem;
            if (tid == 0) {
                const unsigned prev = __hip_atomic_fetch_add(p.ks_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = prev == (unsigned)ksplit - 1;
            }
            __syncthreads();
            if (!*flag) return;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                for (int q = 0; q < EV; ++q) vv[ps][q] = 0.f;
            for (int s2 = 0; s2 < ksplit; ++s2) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                    for (int q = 0; q < EV; q += 4) {
                        const uint4v x = __builtin_amdgcn_raw_buffer_load_b128(
                            rs_part, (int)((((s2 * NPASS + ps) * NT + tid) * EV + q) * sizeof(float)), 0, 16);
                        vv[ps][q] += __uint_as_float(x[0]); vv[ps][q + 1] += __uint_as_float(x[1]);
                        vv[ps][q + 2] += __uint_as_float(x[2]); vv[ps][q + 3] += __uint_as_float(x[3]);
                    }
            }
            if (tid == 0) __hip_atomic_store(p.ks_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
        }
        if (ncol_ok) {
            float bv[EV], osc[EV];
#pragma unroll
            for (int q = 0; q < EV; q += 4) {
                const floatx4 b4 = *(const floatx4 *)(bias + n + q);
                bv[q] = b4[0]; bv[q + 1] = b4[1]; bv[q + 2] = b4[2]; bv[q + 3] = b4[3];
                osc[q] = osc[q + 1] = osc[q + 2] = osc[q + 3] = 1.f;
                if (p.oscale) {                            // (DT_F16X3: the power-of-two row scale of the split pack, undone exactly)
                    const floatx4 s4 = *(const floatx4 *)(p.oscale + (bias - p.bias) + n + q);
                    osc[q] = s4[0]; osc[q + 1] = s4[1]; osc[q + 2] = s4[2]; osc[q + 3] = s4[3];
                }
            }
            T *out = (T *)p.out;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = ps * RPP + r0;
                const int m = m0 + row;
                if (row < BM && m < p.M) {
                    float v[EV];
#pragma unroll
                    for (int q = 0; q < EV; ++q) {
                        float x = vv[ps][q] * osc[q] + bv[q];
                        if (p.res_mode == RES_PRE_RELU) x += rv[ps][q];
                        if (p.relu) x = fmaxf(x, 0.f);
                        if (p.res_mode == RES_POST_RELU) x += rv[ps][q];
                        v[q] = x;
                    }
                    TR::storev(out + (size_t)m * p.Cos + cout_off + n, v);
                    if constexpr (sizeof(T) == 2) {
                        if (p.x3_out > 0) {                    // split output (DT_F16X3): [hi | lo = v - hi]
                            float lo[EV];
#pragma unroll
                            for (int q = 0; q < EV; ++q) lo[q] = v[q] - (float)(_Float16)v[q];
                            TR::storev(out + (size_t)m * p.Cos + cout_off + p.x3_out + n, lo);
                        }
                    }
                }
            }
        }
    } else {
        // NCHW f32: each thread owns FOUR consecutive rows (positions of one channel plane) and walks
        // the columns: 16-byte stores (4 B per lane is store-issue bound: the 63x63 mask logits are
        // 79 MB per B=8 frame batch).  The tensor is handed to the caller and not re-read on the
        // device: streaming (non-temporal) stores.
        typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
        constexpr int RG = BM / 4;           // row groups
        constexpr int CG = NT / RG;          // columns processed concurrently
        const int r4 = (tid % RG) * 4, cg = tid / RG;
        const int m = m0 + r4;
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, pos = m - b * hw;
            const bool vec = (m + 3 < p.M) && (pos + 3 < hw);      // all four rows in one plane
            float *obase = (float *)p.out + (size_t)b * p.N * hw + pos;
            const float *er = ebase + ((r4 >> 6) * WN * WK) * (64 * LDE) + (r4 & 63);
#pragma unroll 2
            for (int j = cg; j < BN; j += CG) {
                const int n = n0 + j;
                if (n < p.N) {
                    const float *ec = er + ((j >> 6) * WK) * (64 * LDE) + (j & 63) * LDE;
                    floatx4 v = *(const floatx4 *)ec;
#pragma unroll
                    for (int q = 1; q < WK; ++q) v += *(const floatx4 *)(ec + q * (64 * LDE));
                    const float bn = bias[n], sn = p.oscale ? p.oscale[(bias - p.bias) + n] : 1.f;
                    v[0] = v[0] * sn + bn; v[1] = v[1] * sn + bn; v[2] = v[2] * sn + bn; v[3] = v[3] * sn + bn;
                    if (p.relu) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    float *o = obase + (size_t)n * hw;
                    if (vec) {
                        float4u vv = {v[0], v[1], v[2], v[3]};
                        if (p.nt_store) __builtin_nontemporal_store(vv, (float4u *)o);
                        else *(float4u *)o = vv;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int mq = m + q;
                            if (mq < p.M) {
                                const int bq = mq / hw, pq = mq - bq * hw;
                                ((float *)p.out)[((size_t)bq * p.N + n) * hw + pq] = v[q];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <typename T, int WM, int WN, int WK, int KT, int OUT_MODE, int NSTAGE>
__global__ __launch_bounds__((SMK_NCW + 4) * 64, (WgPerCu<64 * (WM + WN) * KT, NSTAGE, SMK_EPI, SMK_NCW + 4>::waves_per_simd))
void conv_igemm_kernel(const ConvBatch cb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[IgemmLds<WM, WN, WK, KT, NSTAGE>::v];
    set_wave_prio(cb.p[0].wave_prio);
    conv_igemm_body<T, WM, WN, WK, KT, OUT_MODE, NSTAGE>(cb, (int)blockIdx.x, (int)blockIdx.z, 0, smem);
}

// ---------------------------------------------------------------------------------------------
// conv3x3_halo_kernel<T, WM, WN, WK, AROWS, NSLOT, NPATCH> -- 3x3 stride-1 convolutions with the activation patch
// staged ONCE per channel chunk and shared by all nine taps.
//
// The implicit GEMM above re-stages the A rows for every tap (9 x BM x 128 B per 64-channel chunk).
// Here the K loop runs chunk-major (chunk, kh, kw): for a chunk the producers load the tile's input
// FOOTPRINT -- every padded-input pixel any of its BM output pixels touches, as rows of 128 bytes in
// padded row-major order -- and the consumers read tap (kh, kw) of output row r at LDS row
//     arow[r] + kh*dil*Wp + kw*dil          (arow[r] = oy*Wp + ox - q0, Wp = padded input width)
// i.e. the nine taps are nine constant row offsets into the same patch.  A tile covers BM consecutive
// output pixels of ONE image (tiles do not straddle images), so the footprint is a contiguous range of
// the padded-linear pixel index: BM + a few row wraps + 2*dil*(Wp+1) rows instead of 9*BM.
// Weights need the chunk-major K order (PackedConv::w_halo).  One patch buffer (single: a new chunk
// starts with an extra barrier) + an NSLOT-deep weight ring (counted vmcnt inside a chunk) keep the LDS at
// <= 76 KB, two workgroups per CU.  NPATCH = 2 (launches of at most one workgroup per CU, where the LDS is free
// anyway): the patch is double-buffered -- the next chunk's patch is fetched while the current chunk's nine taps
// run, the extra barrier and the exposed patch latency (~1 us per chunk boundary) disappear.
// NHWC epilogue only.  p.buf_lds must be set (SRD range check supplies all zero padding).
// ---------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int WK, int AROWS, int NSLOT, int NPATCH>
__global__ __launch_bounds__(512, (NPATCH == 2 ? 2 : 4)) void conv3x3_halo_kernel(const ConvParams p) {
    static_assert(NPATCH == 1 || (NPATCH == 2 && NSLOT >= 3), "double-buffered patch needs a weight ring of >= 3");
    static_assert(WM * WN * WK == 4 && (WK == 1 || WK == 2), "four consumers, K split <= 2");
    set_wave_prio(p.wave_prio);
    typedef Traits<T> TR;
    typedef typename TR::frag_t frag_t;
    constexpr int NCW = 4, NT = 512;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int KT = 128;                          // bytes of channels per chunk and per weight K tile
    constexpr int CH = KT / (int)sizeof(T);          // channels per chunk
    constexpr int RB = BN / 32, NRND = AROWS / 32;   // LDS-DMA pieces per producer thread: weights per tap / patch
    constexpr int NKS = 4 / WK;
    constexpr int A_BYTES = AROWS * KT, W_STAGE = BN * KT;
    constexpr int LDE = 68;
    constexpr int EPI_BYTES = NCW * 64 * LDE * 4;
    constexpr int LDS_BYTES = CMax<NPATCH * A_BYTES + NSLOT * W_STAGE, EPI_BYTES>::v;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char *sA = smem, *sW = smem + NPATCH * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *bias = p.bias;
    const int howo = p.Ho * p.Wo, Wp = p.Wl + 2 * p.pad;
    const int tpi = (howo + BM - 1) / BM;            // tiles per image
    const int tilesN = (p.Nst + BN - 1) / BN;
    int t = blockIdx.x;
    if (p.xcd_mode != 0) {                           // XCD-contiguous tm-major order (see the kernel above)
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7;
        const int x = t & 7, j = t >> 3;
        t = x * q + (x < r ? x : r) + j;
    }
    const int tm = t / tilesN, tn = t - tm * tilesN;
    const int b = tm / tpi, ml0 = (tm - b * tpi) * BM;
    const int n0 = tn * BN;
    const int oy0 = ml0 / p.Wo, ox0 = ml0 - oy0 * p.Wo;
    const int q0 = oy0 * Wp + ox0;                   // padded-linear index of output row 0 at tap (0,0)
    const int nvalid = howo - ml0 < BM ? howo - ml0 : BM;
    const int nch = p.Ci / CH, nk = nch * 9;

    floatx16 acc[2][2];
    if (wave >= NCW) {
        // =========================== PRODUCER ===================================================
        const int ptid = tid - NCW * 64, pw = wave - NCW;
        const int lrow = ptid >> 3;
        const int slot = (ptid & 7) ^ swz<8>(lrow);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, p.w_bytes, 0x00020000);
        constexpr unsigned OOB = 0x7ffff000u;
        int org_y = p.org_y, org_x = p.org_x;
        if (p.pos) {
            org_y += p.pos[2 * b + 0] * p.pos_mul + p.pos_add;
            org_x += p.pos[2 * b + 1] * p.pos_mul + p.pos_add;
        }
        const int mlast = ml0 + nvalid - 1;
        const int oyl = mlast / p.Wo, oxl = mlast - oyl * p.Wo;
        const int nrows = oyl * Wp + oxl - q0 + 2 * p.dil * Wp + 2 * p.dil + 1;
        unsigned aoff[NRND];
#pragma unroll
        for (int i = 0; i < NRND; ++i) {
            const int j = lrow + 32 * i;
            const int q = q0 + j;
            const int iyp = q / Wp, ixp = q - iyp * Wp;
            const int ly = iyp - p.pad, lx = ixp - p.pad;
            const int sy = ly + org_y, sx = lx + org_x;
            const bool ok = (j < nrows) & ((unsigned)ly < (unsigned)p.Hl) & ((unsigned)lx < (unsigned)p.Wl) &
                            ((unsigned)sy < (unsigned)p.Hs) & ((unsigned)sx < (unsigned)p.Ws);
            const long off = (((long)(b * p.Hs + sy) * p.Ws + sx) * p.Cs + p.cin_off) * (long)sizeof(T) + slot * 16;
            aoff[i] = ok ? (unsigned)off : OOB;
        }
        unsigned wofs[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) wofs[i] = (unsigned)((size_t)(n0 + lrow + 32 * i) * p.Kpad * sizeof(T)) + slot * 16;
        auto issue_A = [&](int c) {
            unsigned char *d = sA + (NPATCH == 2 ? (c & 1) * A_BYTES : 0) + pw * 1024;
#pragma unroll
            for (int i = 0; i < NRND; ++i)
                if (NPATCH == 2 || i * 32 < nrows)         // (wave-uniform) rounds beyond the footprint are skipped;
                                                           // NPATCH == 2 issues all: its vmcnt counts are static
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_t *)(d + i * 4096), 16,
                                                             (int)(aoff[i] == OOB ? OOB : aoff[i] + (unsigned)c * KT), 0, 0, 0);
        };
        int wslot = 0;                                 // ring slot of the next weight tile to issue
        auto issue_W = [&](int kt) {
            unsigned char *d = sW + wslot * W_STAGE + pw * 1024;
#pragma unroll
            for (int i = 0; i < RB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t *)(d + i * 4096), 16, (int)wofs[i], kt * KT, 0, 0);
            wslot = wslot + 1 == NSLOT ? 0 : wslot + 1;
        };
        issue_A(0);
#pragma unroll
        for (int j = 0; j < NSLOT - 1; ++j)
            if (j < nk) issue_W(j);
        int tap = 0, chunk = 0;                        // tap / chunk index of tile kt
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt landed?  In flight behind it: tiles kt+1 .. kt+NSLOT-2, plus
            //   NPATCH == 1: when kt opens a chunk, the patch, which is the youngest load -> drain everything;
            //   NPATCH == 2: the next chunk's patch, issued right after barrier(9c) and BEFORE tile 9c+NSLOT-1, so it
            //                is younger than tile 9c+1 only (loads complete in order): one step allows NRND more
            if (kt + NSLOT - 2 >= nk || (NPATCH == 1 && tap == 0)) wait_vmcnt<0>();
            else if (NPATCH == 2 && tap == 1 && chunk + 1 < nch) wait_vmcnt<RB * (NSLOT - 2) + NRND>();
            else wait_vmcnt<RB * (NSLOT - 2)>();
            __builtin_amdgcn_s_barrier();              // barrier(kt): tile kt complete, tile kt-1 released
            asm volatile("" ::: "memory");
            if (NPATCH == 2 && tap == 0 && chunk + 1 < nch) issue_A(chunk + 1);   // buffer of chunk-1: released by barrier(kt)
            if (kt + NSLOT - 1 < nk) issue_W(kt + NSLOT - 1);
            if (++tap == 9) {                          // tile kt+1 opens a chunk
                tap = 0;
                ++chunk;
                if (NPATCH == 1 && kt + 1 < nk) {      // single buffer: the patch is still read by tile kt
                    __builtin_amdgcn_s_barrier();      // barrier(x): consumers are done with the patch
                    asm volatile("" ::: "memory");
                    issue_A(chunk);
                }
            }
        }
    } else {
        // =========================== CONSUMER ===================================================
        const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int frow = lane & 31, fhalf = lane >> 5;
        int arow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = ml0 + wm * 64 + i * 32 + frow;
            const int oy = ml / p.Wo, ox = ml - oy * p.Wo;
            arow[i] = ml < howo ? oy * Wp + ox - q0 : 0;
        }
        const int fsw = swz<8>(frow);
        const int b_row_off = (wn * 64 + frow) * KT;
        frag_t fa[2][2], fb[2][2];
        // toff: LDS row offset of the tap, plus the patch buffer's row offset (NPATCH == 2: AROWS rows apart;
        // AROWS is a multiple of 16, so the swizzle term of a row is the same in both buffers)
        auto read_frags = [&](int wso, int toff, int s, frag_t (&a)[2], frag_t (&bq)[2]) {
            const int sl = (s * WK + wk) * 2 + fhalf;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = arow[i] + toff;
                a[i] = *(const frag_t *)(sA + ra * KT + ((sl ^ ((ra >> 1) & 7)) << 4));
            }
            const unsigned char *sb = sW + wso + b_row_off + ((sl ^ fsw) << 4);
            bq[0] = *(const frag_t *)sb;
            bq[1] = *(const frag_t *)(sb + 32 * KT);
        };
        auto mma_part = [&](int par, int q0_, int q1_) {
#pragma unroll
            for (int q = q0_; q < q1_; ++q) TR::mma(acc[q >> 1][q & 1], fa[par][q >> 1], fb[par][q & 1]);
        };
        auto tap_off = [&](int tap) {
            const int kh = (tap * 11) >> 5;            // tap / 3 for tap < 9
            return (kh * Wp + (tap - 3 * kh)) * p.dil;
        };
        __builtin_amdgcn_s_barrier();                  // barrier(0)
        asm volatile("" ::: "memory");
        int tap = 0, toff = 0, wso = 0;                // wso: byte offset of tile kt's weight slot
        int pbuf = 0;                                  // patch buffer row offset of tile kt's chunk (0 / AROWS)
        read_frags(0, 0, 0, fa[0], fb[0]);
        for (int kt = 0; kt < nk; ++kt) {
            int tapn = tap + 1;
            int pbufn = pbuf;
            if (tapn == 9) {
                tapn = 0;
                if (NPATCH == 2) pbufn = AROWS - pbuf;
            }
            const int toffn = tap_off(tapn) + pbufn;
            const int wson = wso + W_STAGE == NSLOT * W_STAGE ? 0 : wso + W_STAGE;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s == 0) {
                    mma_part(0, 0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(wso, toff, 1, fa[1], fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(0, 1, 4);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (s + 1 < NKS) {
                    read_frags(wso, toff, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(s & 1, 0, 4);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    mma_part(s & 1, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 1 < nk) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (NPATCH == 1 && tapn == 0) {
                            __builtin_amdgcn_s_barrier();      // barrier(x): the patch may be replaced
                            asm volatile("" ::: "memory");
                        }
                        __builtin_amdgcn_s_barrier();          // barrier(kt+1)
                        asm volatile("" ::: "memory");
                        read_frags(wson, toffn, 0, fa[0], fb[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(s & 1, 2, 4);
                }
            }
            tap = tapn;
            toff = toffn;
            pbuf = pbufn;
            wso = wson;
        }
    }
    __syncthreads();

    // ---- epilogue (NHWC): accumulators -> LDS -> K-group sum -> bias / residual / ReLU -> 16-byte stores
    constexpr int EV = TR::EV;
    constexpr int LPR = BN / EV, RPP = NT / LPR, NPASS = (BM + RPP - 1) / RPP;
    const int c4 = (tid % LPR) * EV, r0 = tid / LPR;
    const int m0 = b * howo + ml0;
    float rv[NPASS][EV];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int q = 0; q < EV; ++q) rv[ps][q] = 0.f;
    if (p.res_mode != RES_NONE && n0 + c4 < p.Nst) {
        const T *res = (const T *)p.res;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = ps * RPP + r0;
            if (row < nvalid) TR::loadv(res + (size_t)(m0 + row) * p.res_Cs + p.res_coff + n0 + c4, rv[ps]);
        }
    }
    if (wave < NCW) {
        float *e = (float *)smem + wave * (64 * LDE);
        const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    e[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * LDE + j * 32 + frow] = acc[i][j][r];
    }
    __syncthreads();
    const int n = n0 + c4;
    if (n < p.Nst) {
        const float *ebase = (const float *)smem;
        float bv[EV];
#pragma unroll
        for (int q = 0; q < EV; q += 4) {
            const floatx4 b4 = *(const floatx4 *)(bias + n + q);
            bv[q] = b4[0]; bv[q + 1] = b4[1]; bv[q + 2] = b4[2]; bv[q + 3] = b4[3];
        }
        T *out = (T *)p.out;
        const float *ecol = ebase + ((c4 >> 6) * WK) * (64 * LDE) + (c4 & 63);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = ps * RPP + r0;
            if (row < nvalid) {
                const float *er = ecol + ((row >> 6) * WN * WK) * (64 * LDE) + (row & 63) * LDE;
                float v[EV];
#pragma unroll
                for (int q = 0; q < EV; q += 4) {
                    floatx4 x = *(const floatx4 *)(er + q);
#pragma unroll
                    for (int kq = 1; kq < WK; ++kq) x += *(const floatx4 *)(er + kq * (64 * LDE) + q);
                    v[q] = x[0]; v[q + 1] = x[1]; v[q + 2] = x[2]; v[q + 3] = x[3];
                }
#pragma unroll
                for (int q = 0; q < EV; ++q) {
                    float x = v[q] + bv[q];
                    if (p.res_mode == RES_PRE_RELU) x += rv[ps][q];
                    if (p.relu) x = fmaxf(x, 0.f);
                    if (p.res_mode == RES_POST_RELU) x += rv[ps][q];
                    v[q] = x;
                }
                TR::storev(out + (size_t)(m0 + row) * p.Cos + p.cout_off + n, v);
            }
        }
    }
}

// host side: eligibility + launch.  Returns 1 when the geometry does not fit (caller falls back to the
// generic kernel), 0 on success, < 0 on launch errors.  p.wgt must be the chunk-major weight pack.
template <typename T, int WM, int WN, int WK, int AROWS, int NSLOT, int NPATCH>
static int launch_halo_t(const ConvParams &p, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int howo = p.Ho * p.Wo, Wp = p.Wl + 2 * p.pad;
    const int tpi = (howo + BM - 1) / BM;
    int worst = 0;
    for (int tl = 0; tl < tpi; ++tl) {
        const int ml0 = tl * BM, nv = howo - ml0 < BM ? howo - ml0 : BM;
        const int oy0 = ml0 / p.Wo, ox0 = ml0 % p.Wo, ml = ml0 + nv - 1, oyl = ml / p.Wo, oxl = ml % p.Wo;
        const int rows = oyl * Wp + oxl - (oy0 * Wp + ox0) + 2 * p.dil * Wp + 2 * p.dil + 1;
        if (rows > worst) worst = rows;
    }
    if (worst > AROWS) return 1;
    dim3 grid(p.B * tpi * ((p.Nst + BN - 1) / BN));
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, WM, WN, WK, AROWS, NSLOT, NPATCH>), grid, dim3(512), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_conv_halo(const ConvParams &p, int dtype, int bm, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const int ch = dtype == DT_F16 ? 64 : 32;
    if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.stride_x != 1 || p.ups || p.groups > 1 || p.Ci % ch != 0 ||
        p.out_mode != OUT_NHWC || !p.buf_lds || p.Wo != p.Wl + 2 * p.pad - 2 * p.dil)
        return 1;
    // BM = 64 with more than one channel chunk and at most one workgroup per CU: double-buffered patch (104 KB of
    // LDS, which such a launch cannot use otherwise).  g_tune.halo_db = 0 keeps the single-buffered kernel.
    const int ch_n = p.Ci / ch;
    const long wgs64 = (long)p.B * ((p.Ho * p.Wo + 63) / 64) * ((p.Nst + 127) / 128);
    const bool db = bm == 64 && g_tune.halo_db && ch_n >= 2 && wgs64 <= 256;
    if (dtype == DT_F16) {
        if (bm == 128) return launch_halo_t<_Float16, 2, 2, 1, 320, 2, 1>(p, s);
        return db ? launch_halo_t<_Float16, 1, 2, 2, 224, 3, 2>(p, s) : launch_halo_t<_Float16, 1, 2, 2, 224, 3, 1>(p, s);
    }
    if (bm == 128) return launch_halo_t<float, 2, 2, 1, 320, 2, 1>(p, s);
    return db ? launch_halo_t<float, 1, 2, 2, 224, 3, 2>(p, s) : launch_halo_t<float, 1, 2, 2, 224, 3, 1>(p, s);
}

// ---------------------------------------------------------------------------------------------
// chain_mask_kernel: Refine's sequential tail and the 63x63 mask head side by side in ONE launch (horizontal fusion).
// refine_chain_kernel occupies B of the 256 CUs for ~44 us (one workgroup per stream, custom.py:150-153 is a dependent
// chain); the mask head (256 -> 3969 1x1 convolution, custom.py:185, 79 MB of fp32 logits at B = 8) is bound by its
// stores and by nothing on the Refine path -- it only needs head.0's output.  As two launches they cost ~44 + ~40 us
// back to back.  Here workgroups [0, B) run the chain (all 1024 threads) and the others run TWO 128x128 mask-head
// tiles each (threads 0-511 / 512-1023, an LDS half each: the same two-workgroups-per-CU overlap the stand-alone
// launch has; both halves execute the same number of barriers).  Cross-stream forks in the captured graph were the
// measured alternative: hipGraphLaunch then costs ~1 ms of host time per replay.
// ---------------------------------------------------------------------------------------------
}  // namespace smk
namespace smk {
#include "refine_chain_body.inc"
constexpr int CM_CONV_LDS = IgemmLds<2, 2, 1, 128, 2>::v;
constexpr int CM_LDS = CMax<RC_LDS, 2 * CM_CONV_LDS>::v;

__global__ __launch_bounds__(1024) void chain_mask_kernel(const RefineChainParams rp, const ConvBatch cb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[CM_LDS];
    if ((int)blockIdx.x < rp.B) {
        refine_chain_body<false>(rp, (int)blockIdx.x, smem);
    } else {
        const int half = threadIdx.x >> 9;
        const int bx = 2 * ((int)blockIdx.x - rp.B) + half;
        if (bx < cb.start[cb.n])                       // (odd tile count -- refused by the launcher: the last workgroup would run one tile)
            conv_igemm_body<_Float16, 2, 2, 1, 128, OUT_NCHW_F32, 2>(cb, bx, 0, half * 512, smem + half * CM_CONV_LDS);
    }
    if (rp.tail_sem) {
        // pipelined frame step: this launch ends the tail -- the last workgroup to get here lets the next frame's persistent launch
        // through its gate (semaphore V, misc_kernels.hip) instead of a one-thread kernel behind this one.  Only CU ownership hangs
        // on it (every workgroup has issued its last instruction but the exit); the outputs become visible at the launch's end as always.
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(rp.tail_sem + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                __hip_atomic_store(rp.tail_sem + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(rp.tail_sem, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// cb: ONE f16 problem with the NCHW f32 epilogue (the mask head), 128x128 tiles
int launch_chain_mask(const RefineChainParams &rp, ConvBatch &cb, void *stream) {
    if (rp.v2_cs != 32 || rp.v1_cs != 16 || rp.v0_cs != 8 || cb.n != 1) return -1;
    ConvParams &p = cb.p[0];
    if (p.out_mode != OUT_NCHW_F32 || p.groups > 1) return -1;
    const int tiles = ((p.M + 127) / 128) * ((p.Nst + 127) / 128);
    // The two 128x128 tile bodies of a workgroup (threads 0-511 / 512-1023) share workgroup-wide barriers: both halves must
    // run the same number of them, i.e. both must HAVE a tile (same K, ksplit 1).  An odd tile count would leave the last
    // workgroup with one body only -- a path no shape of the network produces (3969 / 128 -> 32 column tiles: always even)
    // and that is therefore refused here instead of relied upon; the caller falls back to the two separate launches.
    if (tiles & 1) return 1;
    cb.start[0] = 0;
    for (int i = 1; i <= CONV_BATCH_MAX; ++i) cb.start[i] = tiles;
    p.ksplit = 1;
    hipLaunchKernelGGL(chain_mask_kernel, dim3(rp.B + (tiles + 1) / 2), dim3(1024), 0, (hipStream_t)stream, rp, cb);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- naive reference kernel: one thread per (m, 4 channels); same params, same packing ----
template <typename T>
__global__ void conv_naive_kernel(const ConvParams p) {
    typedef Traits<T> TR;
    constexpr int VE = TR::VE;
    const int g = blockIdx.z;
    const int cin_off = p.cin_off + g * p.g_cin_off;
    const T *wgt = (const T *)p.wgt + (size_t)g * p.g_wgt_off * p.Kpad;
    const float *bias = p.bias + g * p.g_wgt_off;
    const int cout_off = p.cout_off + g * p.g_cout_off;
    const int nq = p.Nst / 4;
    const long total = (long)p.M * nq;
    const T *in = (const T *)p.in;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / nq), n = (int)(idx - (long)m * nq) * 4;
        const RowInfo r = row_info(p, m, p.pos);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int kvec = 0; kvec < p.K; kvec += VE) {
            const KDecode d = decode_k(kvec, p.Ci, p.kw);
            const long off = gather_offset(p, r, d, cin_off);
            if (off < 0) continue;
            for (int e = 0; e < VE; ++e) {
                const float a = (float)in[off + e];
                for (int j = 0; j < 4; ++j)
                    acc[j] = fmaf(a, (float)wgt[(size_t)(n + j) * p.Kpad + kvec + e], acc[j]);
            }
        }
        if (p.out_mode == OUT_NHWC) {
            floatx4 v = {acc[0] + bias[n], acc[1] + bias[n + 1], acc[2] + bias[n + 2], acc[3] + bias[n + 3]};
            floatx4 rv = {0.f, 0.f, 0.f, 0.f};
            if (p.res_mode != RES_NONE)
                rv = TR::load4((const T *)p.res + (size_t)m * p.res_Cs + p.res_coff + n);
            if (p.res_mode == RES_PRE_RELU) v += rv;
            if (p.relu)
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            if (p.res_mode == RES_POST_RELU) v += rv;
            TR::store4((T *)p.out + (size_t)m * p.Cos + cout_off + n, v);
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, pos = m - b * hw;
            for (int j = 0; j < 4; ++j)
                if (n + j < p.N) {
                    float v = acc[j] + bias[n + j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    ((float *)p.out)[((size_t)b * p.N + n + j) * hw + pos] = v;
                }
        }
    }
}

// ---- dispatch ------------------------------------------------------------------------------
Tuning g_tune;

// Tile configurations: (BM, BN) in {128,64}^2, K tile 128 or 256 bytes, ring depth 2..4.
// 64x64 tiles split K four ways inside the workgroup and therefore need the 256-byte K tile
// (two k-steps per wave per tile).
static int default_kt(int bm, int bn) { return (bm == 64 && bn == 64) ? 256 : 128; }
// 256x128 tiles run eight consumer waves (4x2), 128-byte K tiles only
static int default_stages(int bm, int bn, int kt) {
    if (bm == 128 && bn == 128 && kt == 128) return 2;       // 70 KB with the epilogue: two workgroups per CU
    return (bm + bn) * kt * 3 <= 160 * 1024 ? 3 : 2;
}

// Tile choice, fitted to the per-layer micro-benchmark (profiles/r01_v5_convbench_b{8,64}_f16.json, r01_v4_..._b1):
// the largest tile whose grid still covers the chip about twice (two workgroups per CU hide each
// other's hand-over bubbles), because the bytes staged through LDS per flop fall with the tile
// area and the global->LDS path (~20-25 B/clk/CU beside running MFMAs) is what bounds the loop.
TileChoice choose_tile(const ConvParams &p, int dtype) {
    (void)dtype;
    TileChoice t;
    if (g_tune.force_tile) {
        static const int tb[6][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128}};
        t.bm = tb[g_tune.force_tile][0];
        t.bn = tb[g_tune.force_tile][1];
        t.stages = default_stages(t.bm, t.bn, default_kt(t.bm, t.bn));
    } else {
        const long ng = p.groups > 0 ? p.groups : 1;
        auto tiles = [&](int bm, int bn) {
            return (long)((p.M + bm - 1) / bm) * ((p.Nst + bn - 1) / bn) * ng * 16 / g_tune.min_blocks_x16;
        };
        if (p.Nst <= 64) {
            if (tiles(128, 64) >= 200) { t.bm = 128; t.bn = 64; t.stages = 3; }
            else { t.bm = 64; t.bn = 64; t.stages = 3; }
        } else if (tiles(256, 128) >= 900 && p.K >= 2048) { t.bm = 256; t.bn = 128; t.stages = 3; }
        else if (tiles(128, 128) >= 300) { t.bm = 128; t.bn = 128; t.stages = 2; }
        else if (tiles(64, 128) >= 200) { t.bm = 64; t.bn = 128; t.stages = 3; }
        else { t.bm = 64; t.bn = 64; t.stages = 3; }
    }
    t.kt = g_tune.kt ? g_tune.kt : default_kt(t.bm, t.bn);
    if (t.bm == 64 && t.bn == 64) t.kt = 256;
    if (t.bm == 256) t.kt = 128;
    if (g_tune.stages) t.stages = g_tune.stages;
    return t;
}

// Split-K policy.  The emulation without the reduction (profiles/r01_v6_splitk_probe.txt) promised 2-6 us per
// under-filled launch.  Measured with the real exchange (sc1 stores -> counter -> sc1 loads: three dependent trips
// to device-coherent memory; profiles/r01_v6_ksplit_launch.txt, r01_v6_ksplit_ab.txt):
//   * alone on the chip it pays only for long K on few tiles: Refine's v2.0 (K=4608) 21 -> 14 us (x4) at B=8,
//     21 -> 12 us at B=1; layer3 conv2 at B=1 15 -> 13 us; everything shorter loses;
//   * beside other work the exchange is slow: layer3 conv1 at B=8 10.6 -> 25 (x2) / 42 us (x4), and v2.0 inside
//     its merged launch (830 workgroups, chip full) made the B=8 step 0.915 -> 0.955 ms.
// Hence off by default (g_tune.ksplit = 0; B=1 step 0.537 -> 0.531 ms with auto); auto mode only takes
// K >= 4096 with at most one workgroup per CU after the split and evenly dividing K tiles.
static int pick_ksplit(const ConvParams &p, int bm, int tiles, int nk) {
    const int mode = g_tune.ksplit;                // 0 off, 1 auto, 2 / 4 forced (tests)
    if (!mode || p.groups > 1 || bm > 128) return 1;
    if (mode == 1 && p.K < 4096) return 1;
    for (int sp = 4; sp >= 2; sp >>= 1) {
        if (mode != 1 && sp != mode) continue;
        if (nk % sp == 0 && nk / sp >= (mode == 1 ? 4 : 1) && (mode != 1 || tiles * sp <= 256)) return sp;
    }
    return 1;
}

int conv_ksplit(const ConvParams &p, int dtype, const TileChoice &t) {
    if (p.out_mode != OUT_NHWC || !p.ks_part) return 1;
    const int bk = t.kt / (dtype == DT_F16 ? 2 : 4);
    return pick_ksplit(p, t.bm, ((p.M + t.bm - 1) / t.bm) * ((p.Nst + t.bn - 1) / t.bn), (p.K + bk - 1) / bk);
}

template <typename T, int WM, int WN, int WK, int KT, int OM>
static int launch_stages(ConvBatch &cb, int stages, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int NTHREADS = (WM * WN * WK + 4) * 64;
    constexpr int BK_ = KT / (int)sizeof(T);
    int total = 0, groups = 1;
    size_t part_used = 0;                 // floats / counters of the split-K scratch handed out so far
    int cnt_used = 0;
    constexpr int EVh = Traits<T>::EV, LPRh = BN / EVh, RPPh = NTHREADS / LPRh, NPASSh = (BM + RPPh - 1) / RPPh;
    constexpr size_t PART_PER_SPLIT = (size_t)NPASSh * NTHREADS * EVh;     // f32 per (tile, split)
    for (int i = 0; i < cb.n; ++i) {
        ConvParams &p = cb.p[i];
        cb.start[i] = total;
        const int tiles = ((p.M + BM - 1) / BM) * ((p.Nst + BN - 1) / BN);
        int sp = (OM == OUT_NHWC && p.ks_part) ? pick_ksplit(p, BM, tiles, (p.K + BK_ - 1) / BK_) : 1;
        if (sp > 1 && (part_used + (size_t)tiles * sp * PART_PER_SPLIT > p.ks_part_cap || cnt_used + tiles > p.ks_cnt_cap)) sp = 1;
        p.ksplit = sp;
        if (sp > 1) {
            p.ks_part += part_used;
            p.ks_cnt += cnt_used;
            part_used += (size_t)tiles * sp * PART_PER_SPLIT;
            cnt_used += tiles;
        }
        total += tiles * sp;
        if (p.groups > groups) groups = p.groups;
    }
    for (int i = cb.n; i <= CONV_BATCH_MAX; ++i) cb.start[i] = total;
    dim3 grid(total, 1, groups);
    constexpr int STAGE_BYTES = (BM + BN) * KT;
    constexpr int MAXST = 160 * 1024 / STAGE_BYTES;      // deepest ring that fits the 160 KB LDS
    if (stages > MAXST) stages = MAXST;
    if (stages > 4) stages = 4;
    if (stages < 2) stages = 2;
    if constexpr (MAXST >= 4) {
        if (stages == 4) {
            hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, WK, KT, OM, 4>), grid, dim3(NTHREADS), 0, s, cb);
            return hipGetLastError() == hipSuccess ? 0 : -4;
        }
    }
    if constexpr (MAXST >= 3) {
        if (stages == 3) {
            hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, WK, KT, OM, 3>), grid, dim3(NTHREADS), 0, s, cb);
            return hipGetLastError() == hipSuccess ? 0 : -4;
        }
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, WK, KT, OM, 2>), grid, dim3(NTHREADS), 0, s, cb);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T, int OM>
static int launch_tiles(ConvBatch &cb, TileChoice t, hipStream_t s) {
    const bool k256 = t.kt == 256;
    if (t.bm == 256) return launch_stages<T, 4, 2, 1, 128, OM>(cb, t.stages, s);
    if (t.bm == 128 && t.bn == 128)
        return k256 ? launch_stages<T, 2, 2, 1, 256, OM>(cb, t.stages, s) : launch_stages<T, 2, 2, 1, 128, OM>(cb, t.stages, s);
    if (t.bm == 128 && t.bn == 64)
        return k256 ? launch_stages<T, 2, 1, 2, 256, OM>(cb, t.stages, s) : launch_stages<T, 2, 1, 2, 128, OM>(cb, t.stages, s);
    if (t.bm == 64 && t.bn == 128)
        return k256 ? launch_stages<T, 1, 2, 2, 256, OM>(cb, t.stages, s) : launch_stages<T, 1, 2, 2, 128, OM>(cb, t.stages, s);
    return launch_stages<T, 1, 1, 4, 256, OM>(cb, t.stages, s);
}

// all problems of a batch share dtype, epilogue mode (NHWC / NCHW) and the tile configuration
int launch_conv_mfma_batch(ConvBatch &cb, int dtype, TileChoice t, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (cb.n < 1 || cb.n > CONV_BATCH_MAX) return -1;
    const int om = cb.p[0].out_mode;
    for (int i = 1; i < cb.n; ++i)
        if (cb.p[i].out_mode != om || (cb.p[i].groups > 1) != (cb.p[0].groups > 1)) return -1;
    if (dtype == DT_F16)
        return om == OUT_NHWC ? launch_tiles<_Float16, OUT_NHWC>(cb, t, s) : launch_tiles<_Float16, OUT_NCHW_F32>(cb, t, s);
    return om == OUT_NHWC ? launch_tiles<float, OUT_NHWC>(cb, t, s) : launch_tiles<float, OUT_NCHW_F32>(cb, t, s);
}

int launch_conv_mfma(const ConvParams &p, int dtype, TileChoice t, void *stream) {
    ConvBatch cb;
    cb.n = 1;
    cb.p[0] = p;
    return launch_conv_mfma_batch(cb, dtype, t, stream);
}

int launch_conv_naive(const ConvParams &p, int dtype, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)p.M * (p.Nst / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    if (blocks < 1) blocks = 1;
    dim3 grid(blocks, 1, p.groups > 0 ? p.groups : 1);
    if (dtype == DT_F16) hipLaunchKernelGGL(conv_naive_kernel<_Float16>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_naive_kernel<float>, grid, dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
