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
// Tiling (wave64, 4 waves = 2x2 per workgroup):
//   workgroup tile BM x BN in {64,128}^2, K tile = 128 bytes per row (64 f16 / 32 f32),
//   each wave owns (BM/2)x(BN/2) as 32x32 MFMA fragments:
//     f16: v_mfma_f32_32x32x16_f16   (one per 32 bytes of K)
//     f32: v_mfma_f32_32x32x2_f32    (four per 32 bytes of K; exact fp32 fma chain)
//   A and B tiles are staged global -> VGPR -> LDS (the gather needs per-lane predication),
//   double buffered, one barrier per K tile, loads of tile t+1 in flight under the MFMAs of
//   tile t.  LDS rows are 128 B with the 16-byte slot XOR-swizzled by (row>>1)&7 so that the
//   ds_read_b128 fragment reads are bank-conflict free.  The accumulators go back through
//   LDS once so that global stores (and residual loads) are row-contiguous.
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
};

// byte offset of (row, 16-byte slot) inside a [rows][128 B] LDS tile, XOR swizzled
__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * KTILE_BYTES + ((slot ^ ((row >> 1) & 7)) << 4);
}

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

template <typename T, int BM, int BN, int OUT_MODE, int NSTAGE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    typedef Traits<T> TR;
    constexpr int VE = TR::VE;
    constexpr int BK = KTILE_BYTES / (int)sizeof(T);
    constexpr int FM = BM / 64, FN = BN / 64;     // 32x32 fragments per wave
    constexpr int RA = BM / 32, RB = BN / 32;     // 16-byte LDS-DMA pieces per thread per K tile
    constexpr int WTM = BM / 2, WTN = BN / 2;     // wave tile
    constexpr int LDE = (OUT_MODE == OUT_NCHW_F32) ? WTN + 1 : WTN + 4;
    constexpr int STAGE_BYTES = (BM + BN) * KTILE_BYTES;
    constexpr int EPI_BYTES = 4 * WTM * LDE * 4;
    constexpr int LDS_BYTES = CMax<NSTAGE * STAGE_BYTES, EPI_BYTES>::v;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = blockIdx.z;
    const int cin_off = p.cin_off + g * p.g_cin_off;
    const T *wgt = (const T *)p.wgt + (size_t)g * p.g_wgt_off * p.Kpad;
    const float *bias = p.bias + g * p.g_wgt_off;
    const int cout_off = p.cout_off + g * p.g_cout_off;

    // XCD-aware tile order (p.xcd_mode): workgroup b runs on XCD b % 8 (observed dispatch
    // order, used for speed only).  Mode 1 hands every XCD a contiguous range of the tm-major
    // tile sequence, so the workgroups sharing one activation row panel (same tm, all tn) run
    // on ONE XCD and that panel is fetched into one L2 only.
    const int tilesN = (p.Nst + BN - 1) / BN;
    int t = blockIdx.x;
    if (p.xcd_mode != 0) {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7;
        const int x = t & 7, j = t >> 3;
        t = x * q + (x < r ? x : r) + j;
    }
    int tm, tn;
    if (p.xcd_mode == 2) {            // tn-major: each XCD owns a range of weight panels
        const int tilesM = (p.M + BM - 1) / BM;
        tn = t / tilesM; tm = t - tn * tilesM;
    } else {
        tm = t / tilesN; tn = t - tm * tilesN;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // LDS-DMA writes lane-linear: lane l of wave w fills (row 8w + l/8 [+32i], physical slot l%8).
    // The XOR swizzle therefore goes on the SOURCE: this thread fetches the logical 16-byte
    // slot  phys ^ ((row>>1)&7)  of its row, and the fragment reads apply the same XOR.
    const int lrow = tid >> 3;
    const int slot = (tid & 7) ^ ((lrow >> 1) & 7);

    // ---- per-thread row bookkeeping for the A gather --------------------------------------
    RowInfo ri[RA];
    bool rvalid[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + lrow + 32 * i;
        rvalid[i] = m < p.M;
        ri[i] = row_info(p, rvalid[i] ? m : 0, p.pos);
    }
    const char *in = (const char *)p.in;
    const long zero_off = (const char *)p.zero - in;   // 8 KB of zeros: source of all padding
    const char *wsrc[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        wsrc[i] = (const char *)(wgt + (size_t)(n0 + lrow + 32 * i) * p.Kpad + slot * VE);
    const int nk = (p.K + BK - 1) / BK;

    // This thread always fetches the same 16-byte slot of every K tile, i.e. K index
    // kt*BK + slot*VE.  Its (tap, channel) position advances incrementally; the per-row source
    // offsets are recomputed only when the tap changes (once per Ci/BK tiles on the heavy
    // layers), which keeps the address VALU work out of the MFMA loop.  Padding (conv zero
    // padding, rows >= M, K tail) reads the zero page, so the loads are branch-free.
    KDecode kd = decode_k(slot * VE, p.Ci, p.kw);
    long a_off[RA];                                    // byte offsets relative to `in`
    auto tap_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            long off = -1;
            if (rvalid[i] && kd.kh_i < p.kh) {
                KDecode d0 = kd;
                d0.c = 0;
                off = gather_offset(p, ri[i], d0, cin_off);
            }
            a_off[i] = off >= 0 ? off * (long)sizeof(T) : zero_off;
        }
    };
    tap_offsets();
    auto advance = [&]() {
        kd.c += BK;
        if (kd.c >= p.Ci) {
            do {
                kd.c -= p.Ci;
                if (++kd.kw_i == p.kw) { kd.kw_i = 0; ++kd.kh_i; }
            } while (kd.c >= p.Ci);
            tap_offsets();
        }
    };
    // issue the LDS-DMA of K tile kt into ring slot `buf` (RA + RB loads per thread)
    auto issue = [&](int kt, int buf) {
        unsigned char *sA = smem + buf * STAGE_BYTES + wave * (8 * KTILE_BYTES);
        unsigned char *sB = sA + BM * KTILE_BYTES;
        const long cb = (long)kd.c * (long)sizeof(T);
#pragma unroll
        for (int i = 0; i < RA; ++i) glds16(in + a_off[i] + cb, sA + i * (32 * KTILE_BYTES));
        const long kb = (long)kt * KTILE_BYTES;
#pragma unroll
        for (int i = 0; i < RB; ++i) glds16(wsrc[i] + kb, sB + i * (32 * KTILE_BYTES));
    };

    floatx16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: NSTAGE-1 tiles in flight --------------------------------------------------
    issue(0, 0);
    if (NSTAGE == 3 && nk > 1) { advance(); issue(1, 1); }

    const int frow = lane & 31, fhalf = lane >> 5;
    int cur = 0;                       // ring slot holding tile kt
    for (int kt = 0; kt < nk; ++kt) {
        // my pieces of tile kt have landed (tiles issued after it may still be in flight) ...
        if (NSTAGE == 3 && kt + 1 < nk) wait_vmcnt<RA + RB>();
        else wait_vmcnt<0>();
        // ... and so have everybody else's; all waves are also done reading the slot refilled below
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int ahead = NSTAGE - 1;
        if (kt + ahead < nk) {
            int nxt = cur + ahead;
            if (nxt >= NSTAGE) nxt -= NSTAGE;
            advance();
            issue(kt + ahead, nxt);
        }
        const unsigned char *sA = smem + cur * STAGE_BYTES;
        const unsigned char *sB = sA + BM * KTILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            typename TR::frag_t a[FM], b[FN];
            const int sl = ks * 2 + fhalf;
#pragma unroll
            for (int i = 0; i < FM; ++i)
                a[i] = *(const typename TR::frag_t *)(sA + lds_off(wm * WTM + i * 32 + frow, sl));
#pragma unroll
            for (int j = 0; j < FN; ++j)
                b[j] = *(const typename TR::frag_t *)(sB + lds_off(wn * WTN + j * 32 + frow, sl));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) TR::mma(acc[i][j], a[i], b[j]);
        }
        if (++cur == NSTAGE) cur = 0;
    }
    __syncthreads();

    // ---- epilogue: accumulators -> LDS (per-wave region) -> fused bias/res/relu -> global ---
    float *e = (float *)smem + wave * (WTM * LDE);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                int col = j * 32 + frow;
                e[row * LDE + col] = acc[i][j][r];
            }
    __syncthreads();

    if (OUT_MODE == OUT_NHWC) {
        constexpr int LPR = WTN / 4;      // lanes per output row (4 channels each)
        constexpr int RPP = 64 / LPR;     // rows per pass
        const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
        const int n = n0 + wn * WTN + c4;
        if (n < p.Nst) {
            const floatx4 bv = *(const floatx4 *)(bias + n);
            T *out = (T *)p.out;
            const T *res = (const T *)p.res;
#pragma unroll 4
            for (int pass = 0; pass < WTM / RPP; ++pass) {
                const int row = pass * RPP + r0;
                const int m = m0 + wm * WTM + row;
                if (m < p.M) {
                    floatx4 v = *(const floatx4 *)(e + row * LDE + c4);
                    v += bv;
                    floatx4 rv = {0.f, 0.f, 0.f, 0.f};
                    if (p.res_mode != RES_NONE)
                        rv = TR::load4(res + (size_t)m * p.res_Cs + p.res_coff + n);
                    if (p.res_mode == RES_PRE_RELU) v += rv;
                    if (p.relu) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    if (p.res_mode == RES_POST_RELU) v += rv;
                    TR::store4(out + (size_t)m * p.Cos + cout_off + n, v);
                }
            }
        }
    } else {
        constexpr int CG = 64 / WTM;      // column groups processed concurrently
        const int row = lane % WTM, cg = lane / WTM;
        const int m = m0 + wm * WTM + row;
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, pos = m - b * hw;
            float *obase = (float *)p.out + (size_t)b * p.N * hw + pos;
            for (int j = cg; j < WTN; j += CG) {
                const int n = n0 + wn * WTN + j;
                if (n < p.N) {
                    float v = e[row * LDE + j] + bias[n];
                    if (p.relu) v = fmaxf(v, 0.f);
                    obase[(size_t)n * hw] = v;
                }
            }
        }
    }
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
static int g_num_cu = 256;
Tuning g_tune;

TileChoice choose_tile(const ConvParams &p, int dtype) {
    (void)dtype;
    if (g_tune.force_tile) {
        static const int tb[5][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 128}, {64, 64}};
        TileChoice f{tb[g_tune.force_tile][0], tb[g_tune.force_tile][1]};
        if (p.Nst <= 64) f.bn = 64;
        return f;
    }
    auto blocks = [&](int bm, int bn) {
        return (long)((p.M + bm - 1) / bm) * ((p.Nst + bn - 1) / bn) * (p.groups > 0 ? p.groups : 1);
    };
    TileChoice t;
    t.bn = p.Nst > 64 ? 128 : 64;
    t.bm = p.M > 64 ? 128 : 64;
    // keep every CU busy: shrink the tile while the grid is smaller than the chip
    const long want = (long)g_num_cu * g_tune.min_blocks_x16 / 16;
    if (blocks(t.bm, t.bn) < want && t.bm == 128) t.bm = 64;
    if (blocks(t.bm, t.bn) < want && t.bn == 128) t.bn = 64;
    return t;
}

template <typename T, int BM, int BN, int OM>
static int launch_one(const ConvParams &p, hipStream_t s) {
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Nst + BN - 1) / BN;
    dim3 grid(tilesM * tilesN, 1, p.groups > 0 ? p.groups : 1);
    // ring depth: 128x128 tiles run 2 stages (64 KB -> two workgroups per CU overlap each other),
    // smaller tiles 3 stages (two K tiles in flight per workgroup); measured, see DESIGN.md
    const int stages = g_tune.stages ? g_tune.stages : ((BM == 128 && BN == 128) ? 2 : 3);
    if (stages == 2) hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, OM, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, OM, 3>), grid, dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T, int OM>
static int launch_tiles(const ConvParams &p, TileChoice t, hipStream_t s) {
    if (t.bm == 128 && t.bn == 128) return launch_one<T, 128, 128, OM>(p, s);
    if (t.bm == 128 && t.bn == 64) return launch_one<T, 128, 64, OM>(p, s);
    if (t.bm == 64 && t.bn == 128) return launch_one<T, 64, 128, OM>(p, s);
    return launch_one<T, 64, 64, OM>(p, s);
}

int launch_conv_mfma(const ConvParams &p, int dtype, TileChoice t, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DT_F16) {
        return p.out_mode == OUT_NHWC ? launch_tiles<_Float16, OUT_NHWC>(p, t, s)
                                      : launch_tiles<_Float16, OUT_NCHW_F32>(p, t, s);
    }
    return p.out_mode == OUT_NHWC ? launch_tiles<float, OUT_NHWC>(p, t, s)
                                  : launch_tiles<float, OUT_NCHW_F32>(p, t, s);
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
