// x3_kernels.hip -- the glue kernels of DT_F16X3 contexts (smk_kernels.h): values stored as [hi | lo] fp16 channel planes.
// The convolutions are the fp16 implicit-GEMM kernels on a tripled K (their gathers read the hi plane twice, their epilogues split); what is here is everything
// that is NOT a convolution on the track path: the frame -> split NHWC (tools/test.py:105-112 hands the crop over as float32), the
// stem's max-pool (experiments/siammask_sharp/resnet.py:158), the depth-wise cross-correlation (models/rpn.py:32-38), and the read-back of
// a split tensor for the parity tests.  All arithmetic on hi + lo in fp32; eight channels (one 16-byte vector per plane) per thread, coalesced
// along the channels.
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

namespace {
__device__ __forceinline__ void x3_split(const float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
}  // namespace

// one thread per (pixel, octet of channels): plane-coalesced reads, two 16-byte writes
__global__ __launch_bounds__(256) void cvt_in_x3_kernel(const CvtInParams p) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int oct = p.Cpad >> 3;
    const long hw = (long)p.H * p.W, npix = (long)p.B * hw, total = npix * oct;
    _Float16 *out = (_Float16 *)p.out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pix = idx % npix;                       // pixel-major: the plane reads of a wave coalesce
        const int q = (int)(idx / npix);
        const int b = (int)(pix / hw);
        const long yx = pix - (long)b * hw;
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = q * 8 + e;
            const float v = ch < p.C ? p.in[((size_t)b * p.C + ch) * hw + yx] : 0.f;
            _Float16 h, l;
            x3_split(v, h, l);
            hi[e] = h; lo[e] = l;
        }
        _Float16 *o = out + (size_t)pix * 2 * p.Cpad + q * 8;
        *(half8 *)o = hi; *(half8 *)(o + p.Cpad) = lo;
    }
}

__global__ __launch_bounds__(256) void cvt_out_x3_kernel(const CvtOutParams p) {
    const long hw = (long)p.H * p.W, total = (long)p.B * p.C * hw;
    const _Float16 *in = (const _Float16 *)p.in;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long yx = idx % hw;
        const long t = idx / hw;
        const int c = (int)(t % p.C), b = (int)(t / p.C);
        const _Float16 *s = in + ((size_t)b * hw + yx) * p.Cs + p.coff + c;
        p.out[idx] = (float)s[0] + (float)s[p.plane];
    }
}

// maxpool 3x3 stride 2 pad 1 on a split tensor: the maximum of hi + lo (exact in fp32), split again; eight channels per thread
__global__ __launch_bounds__(256) void maxpool_x3_kernel(const PoolParams p) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int cv = p.C >> 3;
    const long total = (long)p.B * p.Ho * p.Wo * cv;
    const _Float16 *in = (const _Float16 *)p.in;
    _Float16 *out = (_Float16 *)p.out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) << 3;
        long t = idx / cv;
        const int ox = (int)(t % p.Wo); t /= p.Wo;
        const int oy = (int)(t % p.Ho);
        const int b = (int)(t / p.Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)p.W) continue;
                const _Float16 *s = in + ((size_t)(b * p.H + iy) * p.W + ix) * 2 * p.C + c;
                const half8 h = *(const half8 *)s, l = *(const half8 *)(s + p.C);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)h[e] + (float)l[e];
                    m[e] = v > m[e] ? v : m[e];
                }
            }
        }
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h, l;
            x3_split(m[e], h, l);
            hi[e] = h; lo[e] = l;
        }
        _Float16 *o = out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * 2 * p.C + c;
        *(half8 *)o = hi; *(half8 *)(o + p.C) = lo;
    }
}

// depth-wise cross-correlation (models/rpn.py:32-38 conv2d_dw_group) of split tensors: eight channels per thread (16-byte loads of the
// hi and lo planes of both operands), an fp32 FMA chain per channel in (ky, kx) order on hi + lo
__global__ __launch_bounds__(256) void dw_xcorr_x3_kernel(const XcorrParams p) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int cv = p.C >> 3;                               // channel octets computed
    const long total = (long)p.B * p.Ho * p.Wo * cv;
    const _Float16 *x = (const _Float16 *)p.x, *k = (const _Float16 *)p.k;
    _Float16 *out = (_Float16 *)p.out;
    const int Cx = p.Cs;                                   // plane stride of x / k (all branches laid out)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) << 3;
        long t = idx / cv;
        const int ox = (int)(t % p.Wo); t /= p.Wo;
        const int oy = (int)(t % p.Ho);
        const int b = (int)(t / p.Ho);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int ky = 0; ky < p.kh; ++ky)
            for (int kx = 0; kx < p.kw; ++kx) {
                const _Float16 *xs = x + ((size_t)(b * p.H + oy + ky) * p.W + ox + kx) * 2 * Cx + c;
                const _Float16 *ks = k + ((size_t)(b * p.kh + ky) * p.kw + kx) * 2 * Cx + c;
                const half8 xh = *(const half8 *)xs, xl = *(const half8 *)(xs + Cx);
                const half8 kh8 = *(const half8 *)ks, kl = *(const half8 *)(ks + Cx);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = __builtin_fmaf((float)xh[e] + (float)xl[e], (float)kh8[e] + (float)kl[e], acc[e]);
            }
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = (_Float16)acc[e];
            lo[e] = (_Float16)(acc[e] - (float)hi[e]);
        }
        const int g = c >> 8, cc = c & 255;
        _Float16 *o = out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * 2 * Cx + g * 512 + cc;
        *(half8 *)o = hi; *(half8 *)(o + 256) = lo;
    }
}

// the same for the path's 5 x 5 template: one thread = XSEG consecutive outputs of a row x eight channels.  The XSEG + 4 search values of a
// kernel row stay in registers and serve all five kx (28 vector loads per ky for XSEG = 5 outputs instead of 100: the one-output form above
// spends its time re-reading the search row through the L1).  Per output the products are added in the same (ky, kx) order on the same
// fp32 values: bit-identical to dw_xcorr_x3_kernel.
template <int XSEG>
__global__ __launch_bounds__(256) void dw_xcorr_x3_k5_kernel(const XcorrParams p) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    constexpr int KW = 5;
    const int cv = p.C >> 3, nseg = (p.Wo + XSEG - 1) / XSEG;
    const long total = (long)p.B * p.Ho * nseg * cv;
    const _Float16 *x = (const _Float16 *)p.x, *k = (const _Float16 *)p.k;
    _Float16 *out = (_Float16 *)p.out;
    const int Cx = p.Cs;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) << 3;
        long t = idx / cv;
        const int ox0 = (int)(t % nseg) * XSEG; t /= nseg;
        const int oy = (int)(t % p.Ho);
        const int b = (int)(t / p.Ho);
        float acc[XSEG][8];
#pragma unroll
        for (int j = 0; j < XSEG; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
        for (int ky = 0; ky < KW; ++ky) {
            float xf[XSEG + KW - 1][8];
            const _Float16 *xrow = x + ((size_t)(b * p.H + oy + ky) * p.W) * 2 * Cx + c;
#pragma unroll
            for (int j = 0; j < XSEG + KW - 1; ++j) {
                int ix = ox0 + j;
                ix = ix < p.W ? ix : p.W - 1;                  // (columns only a clipped output of the last segment would use)
                const half8 h = *(const half8 *)(xrow + (size_t)ix * 2 * Cx), l = *(const half8 *)(xrow + (size_t)ix * 2 * Cx + Cx);
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[j][e] = (float)h[e] + (float)l[e];
            }
            const _Float16 *krow = k + ((size_t)(b * KW + ky) * KW) * 2 * Cx + c;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const half8 h = *(const half8 *)(krow + (size_t)kx * 2 * Cx), l = *(const half8 *)(krow + (size_t)kx * 2 * Cx + Cx);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float kf = (float)h[e] + (float)l[e];
#pragma unroll
                    for (int j = 0; j < XSEG; ++j) acc[j][e] = __builtin_fmaf(xf[j + kx][e], kf, acc[j][e]);
                }
            }
        }
        const int g = c >> 8, cc = c & 255;
#pragma unroll
        for (int j = 0; j < XSEG; ++j) {
            if (ox0 + j >= p.Wo) break;
            half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = (_Float16)acc[j][e];
                lo[e] = (_Float16)(acc[j][e] - (float)hi[e]);
            }
            _Float16 *o = out + ((size_t)(b * p.Ho + oy) * p.Wo + ox0 + j) * 2 * Cx + g * 512 + cc;
            *(half8 *)o = hi; *(half8 *)(o + 256) = lo;
        }
    }
}

static int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 32768 ? 32768 : b));
}

int launch_cvt_in_x3(const CvtInParams &p, void *stream) {
    if (p.Cpad & 7) return -1;
    hipLaunchKernelGGL(cvt_in_x3_kernel, dim3(grid_for((long)p.B * p.H * p.W * (p.Cpad >> 3))), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_cvt_out_x3(const CvtOutParams &p, void *stream) {
    hipLaunchKernelGGL(cvt_out_x3_kernel, dim3(grid_for((long)p.B * p.C * p.H * p.W)), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_maxpool_x3(const PoolParams &p, void *stream) {
    if (p.C & 7) return -1;
    hipLaunchKernelGGL(maxpool_x3_kernel, dim3(grid_for((long)p.B * p.Ho * p.Wo * (p.C >> 3))), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_xcorr_x3(const XcorrParams &p, void *stream) {
    if ((p.C & 255) || p.Cs < p.C) return -1;
    if (p.kh == 5 && p.kw == 5 && p.W >= 5) {
        constexpr int XSEG = 5;
        hipLaunchKernelGGL(dw_xcorr_x3_k5_kernel<XSEG>, dim3(grid_for((long)p.B * p.Ho * ((p.Wo + XSEG - 1) / XSEG) * (p.C >> 3))), dim3(256), 0,
                           (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    hipLaunchKernelGGL(dw_xcorr_x3_kernel, dim3(grid_for((long)p.B * p.Ho * p.Wo * (p.C >> 3))), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
