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

template <typename T, int BM, int BN, int OUT_MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    typedef Traits<T> TR;
    constexpr int VE = TR::VE;
    constexpr int BK = KTILE_BYTES / (int)sizeof(T);
    constexpr int FM = BM / 64, FN = BN / 64;     // 32x32 fragments per wave
    constexpr int RA = BM / 32, RB = BN / 32;     // 16-byte vectors per thread per K tile
    constexpr int WTM = BM / 2, WTN = BN / 2;     // wave tile
    constexpr int LDE = (OUT_MODE == OUT_NCHW_F32) ? WTN + 1 : WTN + 4;
    constexpr int STAGE_BYTES = (BM + BN) * KTILE_BYTES;
    constexpr int EPI_BYTES = 4 * WTM * LDE * 4;
    constexpr int LDS_BYTES = CMax<2 * STAGE_BYTES, EPI_BYTES>::v;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = blockIdx.z;
    const int cin_off = p.cin_off + g * p.g_cin_off;
    const T *wgt = (const T *)p.wgt + (size_t)g * p.g_wgt_off * p.Kpad;
    const float *bias = p.bias + g * p.g_wgt_off;
    const int cout_off = p.cout_off + g * p.g_cout_off;

    const int tilesN = (p.Nst + BN - 1) / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int slot = tid & 7, lrow = tid >> 3;

    // ---- per-thread row bookkeeping for the A gather --------------------------------------
    RowInfo ri[RA];
    bool rvalid[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + lrow + 32 * i;
        rvalid[i] = m < p.M;
        ri[i] = row_info(p, rvalid[i] ? m : 0, p.pos);
    }
    const T *wrow[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) wrow[i] = wgt + (size_t)(n0 + lrow + 32 * i) * p.Kpad;

    const T *in = (const T *)p.in;
    const int nk = (p.K + BK - 1) / BK;

    uint4 ra[RA], rb[RB];
    auto load_tile = [&](int kt) {
        const int kvec = kt * BK + slot * VE;
        const bool kvalid = kvec < p.K;
        const KDecode d = decode_k(kvec, p.Ci, p.kw);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            long off = (kvalid && rvalid[i]) ? gather_offset(p, ri[i], d, cin_off) : -1;
            if (off >= 0) ra[i] = *(const uint4 *)(in + off);
            else ra[i] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) rb[i] = *(const uint4 *)(wrow[i] + kvec);
    };
    auto store_tile = [&](int buf) {
        unsigned char *sA = smem + buf * STAGE_BYTES;
        unsigned char *sB = sA + BM * KTILE_BYTES;
#pragma unroll
        for (int i = 0; i < RA; ++i) *(uint4 *)(sA + lds_off(lrow + 32 * i, slot)) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(uint4 *)(sB + lds_off(lrow + 32 * i, slot)) = rb[i];
    };

    floatx16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
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
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

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

TileChoice choose_tile(const ConvParams &p, int dtype) {
    (void)dtype;
    auto blocks = [&](int bm, int bn) {
        return (long)((p.M + bm - 1) / bm) * ((p.Nst + bn - 1) / bn) * (p.groups > 0 ? p.groups : 1);
    };
    TileChoice t;
    t.bn = p.Nst > 64 ? 128 : 64;
    t.bm = p.M > 64 ? 128 : 64;
    // keep every CU busy: shrink the tile while the grid is smaller than the chip
    if (blocks(t.bm, t.bn) < g_num_cu && t.bm == 128) t.bm = 64;
    if (blocks(t.bm, t.bn) < g_num_cu && t.bn == 128) t.bn = 64;
    return t;
}

template <typename T, int BM, int BN, int OM>
static int launch_one(const ConvParams &p, hipStream_t s) {
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Nst + BN - 1) / BN;
    dim3 grid(tilesM * tilesN, 1, p.groups > 0 ? p.groups : 1);
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, OM>), grid, dim3(256), 0, s, p);
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
