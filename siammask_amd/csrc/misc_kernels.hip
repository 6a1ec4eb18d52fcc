// misc_kernels.hip -- HBM-bound kernels of the SiamMask path (gfx950):
//   * dw_xcorr   : depth-wise cross-correlation, models/rpn.py:32-38 (conv2d_dw_group)
//   * maxpool    : nn.MaxPool2d(3, 2, 1), experiments/siammask_sharp/resnet.py:158
//   * cvt_in/out : NCHW f32 <-> NHWC dtype at the drop-in boundary
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <typename T> struct P2;   // a pair of channels
template <> struct P2<float> {
    static __device__ inline floatx2 ld(const float *p) { return *(const floatx2 *)p; }
    static __device__ inline void st(float *p, floatx2 v) { *(floatx2 *)p = v; }
};
template <> struct P2<_Float16> {
    static __device__ inline floatx2 ld(const _Float16 *p) {
        half2_t h = *(const half2_t *)p;
        floatx2 v = {(float)h[0], (float)h[1]};
        return v;
    }
    static __device__ inline void st(_Float16 *p, floatx2 v) {
        half2_t h = {(_Float16)v[0], (_Float16)v[1]};
        *(half2_t *)p = h;
    }
};

// ------------------------------------------------------------------------------------------
// dw_xcorr: out[b,i,j,c] = sum_{u,v} x[b,i+u,j+v,c] * k[b,u,v,c]        (NHWC, no flip, valid)
//
// Channels sit on lanes (NHWC), so every dot product is lane-local: no cross-lane reduction
// is needed and every global access is a contiguous 128/256-byte run of 64 channels.
// Workgroup = (band of XC_BR output rows) x (64 channels) x (batch item): the (BR+kh-1) input
// rows of that channel chunk and the kh*kw taps are staged in LDS with 16-byte coalesced loads;
// each thread owns a channel pair and one half-row strip of outputs and slides the kw-wide
// window along x in registers (each staged value is read from LDS once per tap row).
// ------------------------------------------------------------------------------------------
constexpr int XC_BR = 5;       // output rows per workgroup
constexpr int XC_SW = 13;      // max outputs per thread strip (half of a 25-wide row)
constexpr int XC_KMAX = 5;     // max taps per row
constexpr int XC_THREADS = 32 * XC_BR * 2;   // upper bound (64-channel workgroups)

// XC_CH = channels per workgroup (64 or 32): 32 doubles the number of workgroups, whose load / compute /
// store phases then overlap better on a CU (this kernel is latency-, not bandwidth-limited at B=8)
template <typename T, int XC_CH, bool BATCH = false>
__global__ __launch_bounds__(XC_THREADS) void dw_xcorr_kernel(const XcorrParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    constexpr int VE = 16 / (int)sizeof(T);
    constexpr int VPP = XC_CH / VE;                 // 16-byte vectors per pixel chunk
    const int band = blockIdx.x, c0 = blockIdx.y * XC_CH, b = blockIdx.z;
    const int i0 = band * XC_BR;
    const int rows_out = min(XC_BR, p.Ho - i0);
    const int rows_in = rows_out + p.kh - 1;
    T *sx = (T *)xsm;                                // [rows_in][W][64]
    T *sk = sx + (XC_BR + XC_KMAX - 1) * p.W * XC_CH; // [kh*kw][64]
    const T *x = (const T *)p.x;
    const T *k = (const T *)p.k;

    const int nvx = rows_in * p.W * VPP;
    const int nthr = blockDim.x;
    if (BATCH) {
        // (measurement variant, xc_full = 2: all of a thread's 16-byte loads in flight before the first LDS write)
        constexpr int NLD = 8;
        const size_t xb = ((size_t)(b * p.H + i0) * p.W) * p.Cs + c0;       // the band's rows are contiguous pixels
        for (int v0 = threadIdx.x; v0 < nvx; v0 += NLD * nthr) {
            uint4 r[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int v = (v0 + i * nthr < nvx) ? v0 + i * nthr : v0;
                const int pix = v / VPP, q = v - pix * VPP;
                r[i] = *(const uint4 *)(x + xb + (size_t)pix * p.Cs + q * VE);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int v = v0 + i * nthr;
                if (v < nvx) {
                    const int pix = v / VPP, q = v - pix * VPP;
                    *(uint4 *)(sx + (size_t)pix * XC_CH + q * VE) = r[i];
                }
            }
        }
    } else {
        for (int v = threadIdx.x; v < nvx; v += nthr) {
            const int pix = v / VPP, q = v - pix * VPP;
            const int r = pix / p.W, col = pix - r * p.W;
            const size_t g = ((size_t)(b * p.H + i0 + r) * p.W + col) * p.Cs + c0 + q * VE;
            *(uint4 *)(sx + (size_t)pix * XC_CH + q * VE) = *(const uint4 *)(x + g);
        }
    }
    const int nvk = p.kh * p.kw * VPP;
    for (int v = threadIdx.x; v < nvk; v += nthr) {
        const int tap = v / VPP, q = v - tap * VPP;
        const size_t g = ((size_t)b * p.kh * p.kw + tap) * p.Cs + c0 + q * VE;
        *(uint4 *)(sk + (size_t)tap * XC_CH + q * VE) = *(const uint4 *)(k + g);
    }
    __syncthreads();

    constexpr int NCP = XC_CH / 2;                   // channel pairs per workgroup
    const int cp = threadIdx.x % NCP;                // channel pair
    const int strip = threadIdx.x / NCP;             // 0 .. 2*BR-1 (threads beyond that idle when XC_CH = 32 ... see launch)
    const int ri = strip >> 1, half = strip & 1;
    if (ri >= rows_out) return;
    const int wfirst = (p.Wo + 1) / 2;               // 13 of 25
    const int j0 = half ? wfirst : 0;
    const int jn = half ? p.Wo - wfirst : wfirst;    // outputs in this strip (<= XC_SW)

    floatx2 acc[XC_SW];
#pragma unroll
    for (int j = 0; j < XC_SW; ++j) acc[j] = floatx2{0.f, 0.f};

    for (int u = 0; u < p.kh; ++u) {
        floatx2 tap[XC_KMAX];
#pragma unroll
        for (int v = 0; v < XC_KMAX; ++v)
            tap[v] = v < p.kw ? P2<T>::ld(sk + (size_t)(u * p.kw + v) * XC_CH + cp * 2) : floatx2{0.f, 0.f};
        const T *srow = sx + (size_t)((ri + u) * p.W + j0) * XC_CH + cp * 2;
#pragma unroll
        for (int t = 0; t < XC_SW + XC_KMAX - 1; ++t) {
            // input column j0 + t contributes to outputs jj = t - v, v = 0..kw-1
            floatx2 xv = floatx2{0.f, 0.f};
            if (t < jn + p.kw - 1) xv = P2<T>::ld(srow + (size_t)t * XC_CH);
#pragma unroll
            for (int v = 0; v < XC_KMAX; ++v) {
                const int jj = t - v;
                if (jj >= 0 && jj < XC_SW) {
                    acc[jj][0] = fmaf(xv[0], tap[v][0], acc[jj][0]);
                    acc[jj][1] = fmaf(xv[1], tap[v][1], acc[jj][1]);
                }
            }
        }
    }
    T *out = (T *)p.out;
#pragma unroll
    for (int j = 0; j < XC_SW; ++j)
        if (j < jn) {
            const size_t g = ((size_t)(b * p.Ho + i0 + ri) * p.Wo + j0 + j) * p.Cs + c0 + cp * 2;
            P2<T>::st(out + g, acc[j]);
        }
}

// Tall-band variant (round 3): workgroup = (64 channels) x (a band of `band` output rows: 13 of the 25 -> two bands per image).
// The 5-row bands above re-read kh - 1 of every 9 input rows (1.8x the input: 25.3 MB fabric-side for 18.3 MB algorithmic at
// B = 8) and their 480 small workgroups each pay a whole load -> compute -> store latency chain.  With 13-row bands the input
// is read 1.14x (17 + 16 of 29 rows), every access is a full 128-byte line of 64 channels (a 32-channel variant halves the
// LDS footprint but makes every line a shared, partially written one: measured 52 MB fabric-side), 24 x B workgroups of 13
// waves; a thread owns a channel pair and ONE half-row strip, and all of its 16-byte loads are in flight before the first
// LDS write.
template <typename T, int XC_CH>
__global__ __launch_bounds__(1024) void dw_xcorr_tall_kernel(const XcorrParams p, const int band) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    constexpr int VE = 16 / (int)sizeof(T);
    constexpr int VPP = XC_CH / VE;
    const int c0 = blockIdx.x * XC_CH, i0 = blockIdx.y * band, b = blockIdx.z;
    const int rows_out = min(band, p.Ho - i0);
    const int rows_in = rows_out + p.kh - 1;
    T *sx = (T *)xsm;                                // [rows_in][W][XC_CH]
    T *sk = sx + (size_t)(band + p.kh - 1) * p.W * XC_CH;   // [kh*kw][XC_CH]
    const T *x = (const T *)p.x;
    const T *k = (const T *)p.k;
    const int nthr = blockDim.x;
    const int nvk = p.kh * p.kw * VPP;
    for (int v = threadIdx.x; v < nvk; v += nthr) {
        const int tap = v / VPP, q = v - tap * VPP;
        const size_t g = ((size_t)b * p.kh * p.kw + tap) * p.Cs + c0 + q * VE;
        *(uint4 *)(sk + (size_t)tap * XC_CH + q * VE) = *(const uint4 *)(k + g);
    }
    const int nvx = rows_in * p.W * VPP;
    const size_t xb = ((size_t)(b * p.H + i0) * p.W) * p.Cs + c0;      // the band's input rows are contiguous pixels
    constexpr int NLD = 6;
    for (int v0 = threadIdx.x; v0 < nvx; v0 += NLD * nthr) {
        uint4 r[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {                  // (unconditional loads of a clamped index: a load under a branch
            const int v = (v0 + i * nthr < nvx) ? v0 + i * nthr : v0;   // (OOB: re-load the own first vector, an L1 hit; one shared address would serialise)
            const int pix = v / VPP, q = v - pix * VPP;
            r[i] = *(const uint4 *)(x + xb + (size_t)pix * p.Cs + q * VE);
        }
        asm volatile("" ::: "memory");                   // (keeps the compiler from sinking each load to its store)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = v0 + i * nthr;
            if (v < nvx) {
                const int pix = v / VPP, q = v - pix * VPP;
                *(uint4 *)(sx + (size_t)pix * XC_CH + q * VE) = r[i];
            }
        }
    }
    __syncthreads();

    constexpr int NCP = XC_CH / 2;
    const int cp = threadIdx.x % NCP;
    const int strip = threadIdx.x / NCP, nstrip = nthr / NCP;
    const int wfirst = (p.Wo + 1) / 2;
    T *out = (T *)p.out;
    for (int sidx = strip; sidx < 2 * rows_out; sidx += nstrip) {
        const int ri = sidx >> 1, half = sidx & 1;
        const int j0 = half ? wfirst : 0;
        const int jn = half ? p.Wo - wfirst : wfirst;
        floatx2 acc[XC_SW];
#pragma unroll
        for (int j = 0; j < XC_SW; ++j) acc[j] = floatx2{0.f, 0.f};
        for (int u = 0; u < p.kh; ++u) {
            floatx2 tap[XC_KMAX];
#pragma unroll
            for (int v = 0; v < XC_KMAX; ++v)
                tap[v] = v < p.kw ? P2<T>::ld(sk + (size_t)(u * p.kw + v) * XC_CH + cp * 2) : floatx2{0.f, 0.f};
            const T *srow = sx + (size_t)((ri + u) * p.W + j0) * XC_CH + cp * 2;
#pragma unroll
            for (int t = 0; t < XC_SW + XC_KMAX - 1; ++t) {
                floatx2 xv = floatx2{0.f, 0.f};
                if (t < jn + p.kw - 1) xv = P2<T>::ld(srow + (size_t)t * XC_CH);
#pragma unroll
                for (int v = 0; v < XC_KMAX; ++v) {
                    const int jj = t - v;
                    if (jj >= 0 && jj < XC_SW) {
                        acc[jj][0] = fmaf(xv[0], tap[v][0], acc[jj][0]);
                        acc[jj][1] = fmaf(xv[1], tap[v][1], acc[jj][1]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < XC_SW; ++j)
            if (j < jn) {
                const size_t g = ((size_t)(b * p.Ho + i0 + ri) * p.Wo + j0 + j) * p.Cs + c0 + cp * 2;
                P2<T>::st(out + g, acc[j]);
            }
    }
}

// more than 64 KB of dynamic LDS needs an opt-in per kernel; done once per process, outside any stream capture
void xcorr_prepare() {
    static bool done = false;
    if (done) return;
    (void)hipFuncSetAttribute((const void *)dw_xcorr_tall_kernel<_Float16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_xcorr_tall_kernel<float, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    done = true;
}

template <typename T>
static bool launch_xcorr_tall(const XcorrParams &p, hipStream_t s) {
    constexpr int CH = 64;
    if (p.C % CH != 0) return false;
    // as few bands as the LDS (<= 150 KB) and the thread count (one strip per thread, <= 1024) allow
    int nb = 1;
    for (;; ++nb) {
        const int band = (p.Ho + nb - 1) / nb;
        const size_t lds = ((size_t)(band + p.kh - 1) * p.W + (size_t)p.kh * p.kw) * CH * sizeof(T);
        if ((lds <= 150 * 1024 && 2 * band * (CH / 2) <= 1024) || band == 1) break;
    }
    const int band = (p.Ho + nb - 1) / nb;
    const size_t lds = ((size_t)(band + p.kh - 1) * p.W + (size_t)p.kh * p.kw) * CH * sizeof(T);
    if (lds > 150 * 1024) return false;
    xcorr_prepare();                                 // (no-op after the first call; smk_create calls it before any capture)
    int strips = 2 * band;
    if (strips * (CH / 2) > 1024) strips = 1024 / (CH / 2);
    hipLaunchKernelGGL((dw_xcorr_tall_kernel<T, CH>), dim3(p.C / CH, (p.Ho + band - 1) / band, p.B), dim3(strips * (CH / 2)), lds, s, p, band);
    return true;
}

template <typename T, int CH>
static void launch_xcorr_t(const XcorrParams &p, hipStream_t s) {
    const int bands = (p.Ho + XC_BR - 1) / XC_BR;
    dim3 grid(bands, p.C / CH, p.B);
    const size_t lds = ((size_t)(XC_BR + XC_KMAX - 1) * p.W + XC_KMAX * XC_KMAX) * CH * sizeof(T);
    // (CH/2 channel pairs) x (2*BR half-row strips) threads
    if (g_tune.xc_full == 2) hipLaunchKernelGGL((dw_xcorr_kernel<T, CH, true>), grid, dim3((CH / 2) * XC_BR * 2), lds, s, p);
    else hipLaunchKernelGGL((dw_xcorr_kernel<T, CH, false>), grid, dim3((CH / 2) * XC_BR * 2), lds, s, p);
}

int launch_xcorr(const XcorrParams &p, int dtype, void *stream) {
    if (p.kh > XC_KMAX || p.kw > XC_KMAX || (p.Wo + 1) / 2 > XC_SW || p.C % 64 != 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (g_tune.xc_full == 1) {                           // tall bands of 64 channels
        const bool ok = dtype == DT_F16 ? launch_xcorr_tall<_Float16>(p, s) : launch_xcorr_tall<float>(p, s);
        if (ok) return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    const bool c32 = g_tune.xc_ch == 32;
    if (dtype == DT_F16) { if (c32) launch_xcorr_t<_Float16, 32>(p, s); else launch_xcorr_t<_Float16, 64>(p, s); }
    else { if (c32) launch_xcorr_t<float, 32>(p, s); else launch_xcorr_t<float, 64>(p, s); }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ------------------------------------------------------------------------------------------
// maxpool 3x3 stride 2 pad 1 (pads with -inf), NHWC, one 16-byte channel vector per thread
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_kernel(const PoolParams p) {
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const int cv = p.C / VE;
    const long total = (long)p.B * p.Ho * p.Wo * cv;
    const T *in = (const T *)p.in;
    T *out = (T *)p.out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cv);
        long t = idx / cv;
        const int ox = (int)(t % p.Wo); t /= p.Wo;
        const int oy = (int)(t % p.Ho);
        const int b = (int)(t / p.Ho);
        vec_t m;
#pragma unroll
        for (int e = 0; e < VE; ++e) m[e] = (T)(-65504.0f);
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)p.W) continue;
                const vec_t v = *(const vec_t *)(in + ((size_t)(b * p.H + iy) * p.W + ix) * p.C + q * VE);
#pragma unroll
                for (int e = 0; e < VE; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *(vec_t *)(out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.C + q * VE) = m;
    }
}

int launch_maxpool(const PoolParams &p, int dtype, void *stream) {
    const int ve = dtype == DT_F16 ? 8 : 4;
    const long total = (long)p.B * p.Ho * p.Wo * (p.C / ve);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DT_F16) hipLaunchKernelGGL(maxpool_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(maxpool_kernel<float>, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ------------------------------------------------------------------------------------------
// boundary layout conversion: NCHW f32 -> NHWC dtype (channels zero padded to Cpad)
// one thread per (pixel, 8-channel chunk); for the network input (C=3, Cpad=8) that is one
// thread per pixel with plane-coalesced reads and a 16/32-byte write.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void cvt_in_kernel(const CvtInParams p) {
    const int chunks = p.Cpad / 8;
    const long hw = (long)p.H * p.W;
    const long total = (long)p.B * hw * chunks;
    T *out = (T *)p.out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long pix = idx % ((long)p.B * hw);     // pixel-major so that plane reads coalesce
        const int q = (int)(idx / ((long)p.B * hw));
        const int b = (int)(pix / hw);
        const long yx = pix - (long)b * hw;
        T v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = q * 8 + e;
            v[e] = c < p.C ? (T)p.in[((size_t)b * p.C + c) * hw + yx] : (T)0.f;
        }
        T *o = out + (size_t)pix * p.Cpad + q * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e];
    }
}

// stem input: two horizontally adjacent pixels x (3 channels + 1 zero) per 16-byte (f16) vector, so that
// the 7x7 stride-2 stem becomes a 7 x 4 "pixel-pair" convolution with K = 224 instead of 7*7*8 = 392
template <typename T>
__global__ void cvt_in_pairs_kernel(const CvtInParams p) {
    const int wp = (p.W + 1) / 2;
    const long total = (long)p.B * p.H * wp;
    const long hw = (long)p.H * p.W;
    T *out = (T *)p.out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int xp = (int)(idx % wp);
        const long t = idx / wp;
        const int y = (int)(t % p.H), b = (int)(t / p.H);
        T v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int px = 2 * xp + (e >> 2), c = e & 3;
            v[e] = (px < p.W && c < p.C) ? (T)p.in[((size_t)b * p.C + c) * hw + (size_t)y * p.W + px] : (T)0.f;
        }
        T *o = out + (size_t)idx * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e];
    }
}

int launch_cvt_in(const CvtInParams &p, int dtype, void *stream) {
    if (p.pairs) {
        const long total = (long)p.B * p.H * ((p.W + 1) / 2);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 16384) blocks = 16384;
        if (blocks < 1) blocks = 1;
        hipStream_t s = (hipStream_t)stream;
        if (dtype == DT_F16) hipLaunchKernelGGL(cvt_in_pairs_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(cvt_in_pairs_kernel<float>, dim3(blocks), dim3(256), 0, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    const long total = (long)p.B * p.H * p.W * (p.Cpad / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DT_F16) hipLaunchKernelGGL(cvt_in_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(cvt_in_kernel<float>, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// NHWC dtype -> NCHW f32 (tests / debug read-back only)
template <typename T>
__global__ void cvt_out_kernel(const CvtOutParams p) {
    const long hw = (long)p.H * p.W;
    const long total = (long)p.B * p.C * hw;
    const T *in = (const T *)p.in;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long yx = idx % hw;
        const long t = idx / hw;
        const int c = (int)(t % p.C);
        const int b = (int)(t / p.C);
        p.out[idx] = (float)in[((size_t)b * hw + yx) * p.Cs + p.coff + c];
    }
}

int launch_cvt_out(const CvtOutParams &p, int dtype, void *stream) {
    const long total = (long)p.B * p.C * p.H * p.W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DT_F16) hipLaunchKernelGGL(cvt_out_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(cvt_out_kernel<float>, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ------------------------------------------------------------------------------------------
// decode: softmax foreground score, anchor decode, scale/ratio penalty, cosine window, argmax.
// Restates the host code of tools/test.py:205-254 (+ anchors of utils/anchors.py:28-51) per stream WITH THE TOOL'S
// OWN PRECISION STAGES (NumPy >= 2 promotion rules, pinned against the unchanged tool by
// tests/test_decode_reference.py + tests/test_gpu_e2e.py::test_device_decode_*):
//   float32: softmax over the two classes (torch, :206), delta*anchor+anchor, exp(delta)*anchor (:209-212; the anchor
//            table is float32, utils/anchors.py:29), sz(w,h) and w/h (:217-220, :231-232)
//   float64: from the first division by an np.float64 scalar on -- s_c, r_c, penalty, pscore, the window blend (:231-238)
// Every operation is a separate IEEE rounding like the NumPy expressions (explicit *_rn intrinsics: no FMA
// contraction).  The transcendental functions are not bit-identical between libraries (torch's vectorised float32
// exp, NumPy's SIMD exp, this device's exp): float32 exp is computed here as round_to_float(exp(double)), i.e.
// correctly rounded, which sits within 1 ulp of either host library.  Ties resolve to the lowest index like np.argmax.
// Removes the device->host round trip between track_mask and track_refine.
//
// The exp / sqrt / divide chain of one candidate is a long dependent sequence, so the 3125 candidates
// of a stream are spread over A workgroups (one per anchor shape, one candidate per thread).  Each
// workgroup's winner publishes its score, index and finished box; the workgroup that arrives last
// (per-stream counter) picks among the A winners and writes the outputs -- no second launch, no
// recomputation.  The counter is left at zero for the next launch / graph replay.
// ------------------------------------------------------------------------------------------
constexpr int DEC_THREADS = 640;                 // >= S*S = 625 candidates of one anchor shape
__device__ __forceinline__ float exp_f32_cr(float x) { return (float)exp((double)x); }

// one candidate per thread: `have` = this thread owns candidate rem of anchor shape a; c0, c1 = the two class logits,
// l0..l3 = the four box regressions of that candidate (float32, as the network returned them)
__device__ __forceinline__ void decode_tail(const DecodeParams &p, const int a, const int b, const int rem, const bool have,
                                            const float c0, const float c1, const float l0, const float l1, const float l2,
                                            const float l3) {
    const int SS = p.S * p.S;
    // target_sz_in_crop (float64, :230) and the two float64 scalars derived from it
    const double tw = p.target_wh[2 * b], th = p.target_wh[2 * b + 1];
    const double tpad = __dmul_rn(__dadd_rn(tw, th), 0.5);
    const double tsz = __dsqrt_rn(__dmul_rn(__dadd_rn(tw, tpad), __dadd_rn(th, tpad)));
    const double tratio = __ddiv_rn(tw, th);
    const double one_minus_wi = 1.0 - p.window_influence;
    const int ori = -(p.S / 2) * p.stride;
    double best = -1e300;
    int best_i = 0x7fffffff;
    double box[6] = {0., 0., 0., 0., 0., 0.};
    if (have) {
        // float32 stage
        const float m = fmaxf(c0, c1);
        const float e0 = exp_f32_cr(__fsub_rn(c0, m)), e1 = exp_f32_cr(__fsub_rn(c1, m));
        const float score = __fdiv_rn(e1, __fadd_rn(e0, e1));                     // softmax(...)[:, 1]
        const float aw = p.anchor_w[a], ah = p.anchor_h[a];
        const int y = rem / p.S, x = rem - y * p.S;
        const float cx = __fadd_rn(__fmul_rn(l0, aw), (float)(ori + p.stride * x));
        const float cy = __fadd_rn(__fmul_rn(l1, ah), (float)(ori + p.stride * y));
        const float w = __fmul_rn(exp_f32_cr(l2), aw);
        const float h = __fmul_rn(exp_f32_cr(l3), ah);
        const float pad = __fmul_rn(__fadd_rn(w, h), 0.5f);
        const float sz = __fsqrt_rn(__fmul_rn(__fadd_rn(w, pad), __fadd_rn(h, pad)));
        const float ratio = __fdiv_rn(w, h);
        // float64 stage
        double s_c = __ddiv_rn((double)sz, tsz);
        s_c = fmax(s_c, __ddiv_rn(1.0, s_c));
        double r_c = __ddiv_rn(tratio, (double)ratio);
        r_c = fmax(r_c, __ddiv_rn(1.0, r_c));
        const double penalty = exp(__dmul_rn(-__dsub_rn(__dmul_rn(r_c, s_c), 1.0), p.penalty_k));
        const double pscore = __dmul_rn(penalty, (double)score);
        best = __dadd_rn(__dmul_rn(pscore, one_minus_wi), __dmul_rn(p.window[rem], p.window_influence));
        if (!(best == best)) best = -1e300;          // NaN candidates never win (np.argmax would pick the first NaN;
                                                     // the tool's exp overflow case is not a tracked state worth keeping)
        best_i = a * SS + rem;
        box[0] = cx; box[1] = cy; box[2] = w; box[3] = h; box[4] = score; box[5] = penalty;
    }
    // ---- the workgroup's winner: butterfly inside each wave (value, then lowest index), ten partials through LDS (round 6: the first version
    //      walked a 640-entry LDS tree -- ten barriers -- and its last arriver finished the frame on ONE lane, a chain of dependent round trips)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double wb = best;
    int wi = best_i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(wb, off);
        const int oi = __shfl_xor(wi, off);
        if (ob > wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
    }
    __shared__ double sv[DEC_THREADS / 64];
    __shared__ int si[DEC_THREADS / 64];
    __shared__ int s_last, s_row;
    if (lane == 0) { sv[wv] = wb; si[wv] = wi; }
    if (threadIdx.x == 0 && p.ring_box && p.box_out)        // the ring's row: the cursor was advanced by the previous frame's last launch (a plain read, early)
        s_row = (int)(*(volatile const unsigned *)p.ring_cursor % (unsigned)p.ring_rows);   // unsigned: no negative row after 2^31 frames
    __syncthreads();
    double gb = sv[0];
    int gi = si[0];
#pragma unroll
    for (int w2 = 1; w2 < DEC_THREADS / 64; ++w2)
        if (sv[w2] > gb || (sv[w2] == gb && si[w2] < gi)) { gb = sv[w2]; gi = si[w2]; }
    if (best_i == gi) {                              // this workgroup's winner (indices are unique)
        // device-coherent (sc1) stores + vmcnt(0) instead of a release fence: an agent-scope fence writes back
        // and invalidates the XCD's whole L2.  One 64-byte record per (stream, anchor shape): box[0..5], best, best_i.
        unsigned long long *pb = (unsigned long long *)(p.part_box + ((size_t)b * 8 + a) * 8);
        for (int q = 0; q < 6; ++q)
            __hip_atomic_store(pb + q, (unsigned long long)__double_as_longlong(box[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pb + 6, (unsigned long long)__double_as_longlong(best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pb + 7, (unsigned long long)__double_as_longlong((double)best_i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // published before announcing arrival
        const unsigned prev = __hip_atomic_fetch_add(p.arrived + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == (unsigned)p.A - 1;
    }
    __syncthreads();
    if (!s_last || wv != 0) return;
    // ---- last workgroup of this stream, its first WAVE: lane l loads element l & 7 of record l >> 3 -- all A records in ONE round trip (device-
    //      coherent loads) -- then every lane picks the winner from the shuffled (value, index) pairs and lanes 0..7 write the row
    const int rq = lane >> 3, re = lane & 7;
    double rec = 0.0;
    if (rq < p.A) {
        const unsigned long long raw = __hip_atomic_load((const unsigned long long *)(p.part_box + ((size_t)b * 8 + rq) * 8) + re, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
        rec = __longlong_as_double((long long)raw);
    }
    double bv = -1e300;
    int bi = 0x7fffffff, ba = 0;
    for (int q = 0; q < p.A; ++q) {
        const double v = __shfl(rec, 8 * q + 6);
        const int j = (int)__shfl(rec, 8 * q + 7);
        if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; ba = q; }
    }
    const double mine = __shfl(rec, 8 * ba + (lane & 7));          // lane l < 8: element l of the winning record
    const int rm = bi - (bi / SS) * SS, y = rm / p.S, x = rm - y * p.S;
    if (lane == 0 && p.pos_out) {
        // (device-coherent stores: with p.mark set, the Refine tail of a pipelined step reads the position from the side stream as soon
        //  as the mark below is visible -- before this kernel's end has written the XCD's L2 back)
        __hip_atomic_store(p.pos_out + 2 * b, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.pos_out + 2 * b + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane < 8 && p.box_out) {
        p.box_out[8 * b + lane] = mine;
        // result ring: this frame's box also goes to the ring's current row
        if (p.ring_box) p.ring_box[((size_t)s_row * p.B + b) * 8 + lane] = mine;
    }
    if (lane != 0) return;
    if (p.ring_box && p.box_out && p.ring_advance) {
        // Without a Refine launch behind it the last stream to get here advances the cursor -- every other stream's writer has read it
        // (at its start) before its own arrival.
        const unsigned prev = __hip_atomic_fetch_add(p.ring_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (unsigned)p.B - 1) {
            __hip_atomic_store(p.ring_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.ring_cursor, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.ring_also) __hip_atomic_fetch_add(p.ring_also, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __hip_atomic_store(p.arrived + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.mark) {
        // pipelined step: the LAST stream's writer tells the side stream's gate that every position of this frame is in memory
        // (semaphore V; pipe_tail_gate_kernel is the P) -- instead of a one-thread kernel behind this one, 4 us in front of the next
        // frame's first launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(p.mark_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (unsigned)p.B - 1) {
            __hip_atomic_store(p.mark_arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.mark, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


__global__ __launch_bounds__(DEC_THREADS) void decode_kernel(const DecodeParams p) {
    const int a = blockIdx.x, b = blockIdx.y, SS = p.S * p.S;
    const float *cls = p.cls + (size_t)b * 2 * p.A * SS;
    const float *loc = p.loc + (size_t)b * 4 * p.A * SS;
    const int rem = threadIdx.x;
    const bool have = rem < SS;
    float c0 = 0.f, c1 = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    if (have) {
        c0 = cls[a * SS + rem]; c1 = cls[(p.A + a) * SS + rem];
        l0 = loc[(0 * p.A + a) * SS + rem]; l1 = loc[(1 * p.A + a) * SS + rem];
        l2 = loc[(2 * p.A + a) * SS + rem]; l3 = loc[(3 * p.A + a) * SS + rem];
    }
    decode_tail(p, a, b, rem, have, c0, c1, l0, l1, l2, l3);
}

// ---- result ring (smk_set_result_ring): what a tracker keeps per stream and frame for the end-of-batch gather --------------
// (tools/test.py:296-311 keeps the box and the mask per frame; SURVEY.md 8e gathers them at the end of a batch of frames.)
// One launch at the end of the frame step: the decoded box [B][8] f64 and the Refine logits [B][n] f32 -> f16 go to row
// (cursor % rows) of the caller's rings, then the LAST workgroup to finish advances the cursor.  Every workgroup reads the
// cursor before it can possibly have been advanced (the advance needs every workgroup's arrival), and the counter returns to
// zero for the next graph replay.  16-byte loads, 8-byte stores; 0.8 MB at B = 8.
__global__ __launch_bounds__(256) void ring_commit_kernel(const RingParams p) {
    __shared__ int row_sh;
    if (threadIdx.x == 0) row_sh = (int)(*(volatile const unsigned *)p.cursor % (unsigned)p.rows);      // (written by the previous step's launch: visible across the kernel boundary)
    __syncthreads();
    const size_t row = (size_t)row_sh;
    const size_t nref = (size_t)p.B * p.n;
    if (p.ref_ring) {
        _Float16 *dst = p.ref_ring + row * nref;
        const bool vec = ((row * nref) & 3) == 0;                          // 8-byte stores need the row to start on one (B * n may be odd)
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nref; i += (size_t)gridDim.x * 256 * 4) {
            if (vec && i + 4 <= nref) {
                const float4 v = *(const float4 *)(p.ref + i);          // (B * n is odd in general: rows are not 16-byte multiples;
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));  //  i is a multiple of 4 from the base, both bases are 256-byte aligned)
                h4 o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                *(h4 *)(dst + i) = o;
            } else {
                for (size_t j = i; j < i + 4 && j < nref; ++j) dst[j] = (_Float16)p.ref[j];
            }
        }
    }
    if (blockIdx.x == 0 && p.box_ring)
        for (int i = threadIdx.x; i < p.B * 8; i += 256) p.box_ring[row * p.B * 8 + i] = p.box[i];
    // (no release fence: the rows become visible to later kernels at the end of this launch like any other output; the counter
    //  below only orders the cursor's advance behind every workgroup's READ of it, which happened at the top)
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.cursor, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- pipelined frame steps (smk_set_pipeline): the joins between the step's stream and the side stream's Refine / mask tail ----------
// A cross-queue event wait costs 10-22 us on this platform and an event record between two kernels of one stream 4-8 us
// (profiles/r05a_pipelined_timeline_event_join.txt, r05c_order_probe.txt) -- more than half of what the overlap saves.  Both joins are
// therefore one-wave GATE kernels that poll device counters, replayable from captured graphs (baked arguments).  While the main gate
// polls, everything in front of it in its stream has drained and the tail never waits for anything behind it; while the tail's gate
// polls, it holds one wave slot beside the next frame's kernels -- no deadlock.  A partner that does not arrive within 0.2 s raises the
// sequence failure flag (code 3): the caller re-submits, as for a barrier time-out.
// Both joins are SEMAPHORES with one consumer each: P = poll until the count is positive, then take one; V = add one.
//   sem_tail (cnt[0], starts at 1): V by the tail's end (pipe_done_kernel, or chain_mask_kernel's last workgroup; depth 2: by the end of the
//       tail's FIRST part), P by the gate in front of the next frame's persistent launch (outside its batches: in front of the heads);
//   sem_main (cnt[2], starts at 0): V by the decode launch's last writer, P by the gate at the head of the tail;
//   sem_seq  (cnt[7], starts at 0; depth 2): V by conv_seq_kernel's last leaving team, P by the gate at the head of the tail's second part.
// `limit` (100 MHz ticks): the MAIN gate waits for a tail that was enqueued on a free side stream a frame ago -- bounded by the tail's own
// run time, 0.2 s is generous.  A gate at the head of a TAIL starts polling as soon as the side stream is free, i.e. possibly long before
// the step it belongs to even starts on the caller's stream (whatever the caller enqueued in front of the step runs first): its limit is
// 5 s -- and (round 6) those 5 s only start once the step's MAIN part is known to be running: the main gate raises `started`
// (cnt[8]) when it has passed, the tail's gate arms its clock when it sees that and lowers the word again when it passes (the next main gate
// cannot raise it earlier: it waits for this tail's END).  A caller's stream that is blocked in front of the step for longer than 5 s -- a
// collective waiting for a straggler rank, a host-fed event, a large copy -- therefore no longer makes the gate give the frame up; an unarmed
// gate still gives up after two minutes, so that a lost partner never hangs the device for good.
__device__ __forceinline__ void pipe_sem_p(unsigned *sem, int *err, int *err_host, unsigned long long limit, unsigned *started_wait = nullptr,
                                           unsigned *started_set = nullptr) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool armed = started_wait == nullptr;
    while (__hip_atomic_load(sem, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(8);
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (!armed && __hip_atomic_load(started_wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            armed = true;
            t0 = now;
        }
        if (now - t0 > (armed ? limit : 12000000000ull)) {                   // (unarmed: 120 s at 100 MHz)
            __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(err_host, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;                                                          // (the counts are left alone: the host resets them)
        }
    }
    __hip_atomic_fetch_sub(sem, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (started_wait) __hip_atomic_store(started_wait, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (started_set) __hip_atomic_store(started_set, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave, 8 VGPRs, no LDS: a gate that polls THROUGH another stream's kernels (the tail's gate is resident while the next frame's
// persistent launch runs) must fit beside a conv_seq_kernel workgroup -- 2 x 248 of a SIMD's 512 VGPRs.  If it ever did not, the
// persistent launch would not get its CU, decode would never run, and the gates' 0.2 s limits raise the failure flag (loud, no hang).
__global__ __launch_bounds__(64) void pipe_gate_kernel(unsigned *sem, int *err, int *err_host, unsigned long long limit, unsigned *started_wait,
                                                        unsigned *started_set) {
    if (threadIdx.x == 0) pipe_sem_p(sem, err, err_host, limit, started_wait, started_set);
}
__global__ __launch_bounds__(64) void pipe_done_kernel(unsigned *sem) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sem, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// (smk_tune pipe_sig = 1: the main part's completion mark in SIGNAL memory for a hipStreamWaitValue32 on the side stream -- 3-4 us in
//  the two-kernel probe, +50 % per step in the real loop: kept as the measured alternative, profiles/r05e_pipe_sig_ab.txt)
__global__ __launch_bounds__(64) void pipe_mark_kernel(unsigned *sig) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_pipe_mark(unsigned *sig, void *stream) {
    hipLaunchKernelGGL(pipe_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sig);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_pipe_gate(unsigned *sem, int *err, int *err_host, void *stream, int long_wait, unsigned *started_wait, unsigned *started_set) {
    hipLaunchKernelGGL(pipe_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sem, err, err_host,
                       long_wait ? 500000000ull : 20000000ull, started_wait, started_set);             // 5 s / 0.2 s at 100 MHz
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_pipe_done(unsigned *sem, void *stream) {
    hipLaunchKernelGGL(pipe_done_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sem);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ring_commit(const RingParams &p, void *stream) {
    if (!p.cursor || !p.done || p.rows < 1 || p.B < 1) return -1;
    const size_t nref = (size_t)p.B * p.n;
    int grid = (int)((nref + 256 * 4 * 2 - 1) / (256 * 4 * 2));           // two 16-byte loads per thread
    if (grid < 1) grid = 1;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(ring_commit_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_decode(const DecodeParams &p, void *stream) {
    if (p.A > 8 || p.B < 1 || p.S * p.S > DEC_THREADS || !p.part_val || !p.part_idx || !p.part_box || !p.arrived) return -1;
    hipLaunchKernelGGL(decode_kernel, dim3(p.A, p.B), dim3(DEC_THREADS), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
