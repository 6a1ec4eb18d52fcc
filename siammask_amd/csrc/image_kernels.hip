// image_kernels.hip -- the image ops either side of the network, gfx950:
//   * crop_resize : tools/test.py:67-110 get_subwindow_tracking -- crop a square window of the
//     frame, mean-colour padding outside the frame, cv2.resize(INTER_LINEAR) on uint8
//     (OpenCV 3.4 resize.cpp: 11-bit fixed-point taps; exact 2x decimation = 2x2 box average),
//     NCHW f32 out (what im_to_torch hands to Custom.template / track).
//   * paste_mask  : tools/test.py:257-284 -- sigmoid of the 127x127 refine logits, crop_back =
//     cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT -1) into the frame (imgwarp.cpp: inverse map in
//     10-bit fixed point, 1/32-pixel bilinear table, float taps), threshold -> uint8.
// Both are HBM-bound element-wise gathers: one thread per output pixel, coalesced along x.
// Double-precision steps use the non-contracting __d*_rn intrinsics so that the fixed-point
// coordinates are bit-identical to the host restatement in oracle/cv_ops.py.
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

struct LinCoef { int s0, s1, a0, a1; };

// resize.cpp (INTER_LINEAR): source index and the two 11-bit weights of destination index d
__device__ __forceinline__ LinCoef lin_coef(int d, int ssize, double scale) {
    const double t = __dsub_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), 0.5);
    float f = (float)t;
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    LinCoef c;
    c.s0 = s;
    c.s1 = min(s + 1, ssize - 1);
    c.a0 = (int)__builtin_rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    c.a1 = (int)__builtin_rintf(__fmul_rn(f, 2048.f));
    return c;
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const CropParams p) {
    const int b = blockIdx.z;
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= p.model_sz) return;
    const int xmin = p.box[b][0], ymin = p.box[b][1], sz = p.box[b][2];
    const unsigned char *im = p.frames + (size_t)b * p.frame_stride;
    const int H = p.H, W = p.W;
    // patch pixel (py, px) -> frame pixel or the mean colour (tools/test.py:89-100)
    auto px3 = [&](int py, int px, int (&v)[3]) {
        const int y = ymin + py, x = xmin + px;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const unsigned char *q = im + ((size_t)y * W + x) * 3;
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
        } else {
            v[0] = p.avg[b][0]; v[1] = p.avg[b][1]; v[2] = p.avg[b][2];
        }
    };
    int o[3];
    if (sz == p.model_sz) {
        px3(dy, dx, o);
    } else if (sz == 2 * p.model_sz) {
        int a[3], c[3], d[3], e[3];
        px3(2 * dy, 2 * dx, a); px3(2 * dy, 2 * dx + 1, c); px3(2 * dy + 1, 2 * dx, d); px3(2 * dy + 1, 2 * dx + 1, e);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (a[k] + c[k] + d[k] + e[k] + 2) >> 2;
    } else {
        const double scale = __ddiv_rn(1.0, __ddiv_rn((double)p.model_sz, (double)sz));
        const LinCoef cx = lin_coef(dx, sz, scale), cy = lin_coef(dy, sz, scale);
        int p00[3], p01[3], p10[3], p11[3];
        px3(cy.s0, cx.s0, p00); px3(cy.s0, cx.s1, p01); px3(cy.s1, cx.s0, p10); px3(cy.s1, cx.s1, p11);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int h0 = p00[k] * cx.a0 + p01[k] * cx.a1;     // HResizeLinear
            const int h1 = p10[k] * cx.a0 + p11[k] * cx.a1;
            // VResizeLinear, 8-bit specialisation
            int v = (((cy.a0 * (h0 >> 4)) >> 16) + ((cy.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            o[k] = min(max(v, 0), 255);
        }
    }
    const size_t plane = (size_t)p.model_sz * p.model_sz;
    float *out = p.out + (size_t)b * 3 * plane + (size_t)dy * p.model_sz + dx;
    out[0] = (float)o[0];
    out[plane] = (float)o[1];
    out[2 * plane] = (float)o[2];
}

int launch_crop_resize(const CropParams &p, int B, void *stream) {
    if (B < 1 || B > CROP_MAX_B) return -1;
    dim3 grid((p.model_sz + 255) / 256, p.model_sz, B);
    hipLaunchKernelGGL(crop_resize_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ------------------------------------------------------------------------------------------
// warped probability of stream b at frame pixel (x, y): WarpAffineInvoker + remapBilinear<float>
__device__ __forceinline__ float warped_prob(const PasteParams &p, int b, int x, int y) {
    const double *M = p.inv_map[b];
    constexpr int AB_BITS = 10, INTER_BITS = 5, TAB = 1 << INTER_BITS;
    constexpr double AB_SCALE = 1024.0;
    // X = (X0(y) + adelta(x)) >> (AB_BITS - INTER_BITS), round_delta = AB_SCALE / TAB / 2 = 16
    const long adelta = (long)__builtin_rint(__dmul_rn(__dmul_rn(M[0], (double)x), AB_SCALE));
    const long bdelta = (long)__builtin_rint(__dmul_rn(__dmul_rn(M[3], (double)x), AB_SCALE));
    const long X0 = (long)__builtin_rint(__dmul_rn(__dadd_rn(__dmul_rn(M[1], (double)y), M[2]), AB_SCALE)) + 16;
    const long Y0 = (long)__builtin_rint(__dmul_rn(__dadd_rn(__dmul_rn(M[4], (double)y), M[5]), AB_SCALE)) + 16;
    const long X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
    long sxl = X >> INTER_BITS, syl = Y >> INTER_BITS;
    sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);      // saturate_cast<short>
    syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
    const int sx = (int)sxl, sy = (int)syl;
    const float fx = __fmul_rn((float)(X & (TAB - 1)), 1.f / TAB), fy = __fmul_rn((float)(Y & (TAB - 1)), 1.f / TAB);
    const float w00 = __fmul_rn(__fsub_rn(1.f, fy), __fsub_rn(1.f, fx)), w01 = __fmul_rn(__fsub_rn(1.f, fy), fx);
    const float w10 = __fmul_rn(fy, __fsub_rn(1.f, fx)), w11 = __fmul_rn(fy, fx);
    const float *lg = p.logits + (size_t)b * p.ms * p.ms;
    auto tap = [&](int yy, int xx) -> float {
        if ((unsigned)yy < (unsigned)p.ms && (unsigned)xx < (unsigned)p.ms) {
            const float v = lg[yy * p.ms + xx];
            return __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v)));        // .sigmoid() (tools/test.py:256)
        }
        return p.border;
    };
    // remapBilinear<float>: sum of four float products, left to right
    float v = __fmul_rn(tap(sy, sx), w00);
    v = __fadd_rn(v, __fmul_rn(tap(sy, sx + 1), w01));
    v = __fadd_rn(v, __fmul_rn(tap(sy + 1, sx), w10));
    v = __fadd_rn(v, __fmul_rn(tap(sy + 1, sx + 1), w11));
    return v;
}

__global__ __launch_bounds__(256) void paste_mask_kernel(const PasteParams p) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.W) return;
    const float v = warped_prob(p, b, x, y);
    const size_t o = ((size_t)b * p.H + y) * p.W + x;
    if (p.prob_out) p.prob_out[o] = v;
    if (p.mask_out) p.mask_out[o] = v > p.seg_thr ? 1 : 0;
}

// multi-object fusion (tools/test.py:521-523): label = (argmax_o prob_o + 1) * (max_o prob_o > thr);
// np.argmax keeps the first maximum
__global__ __launch_bounds__(256) void paste_labels_kernel(const PasteParams p, int n_obj) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.W) return;
    float best = warped_prob(p, 0, x, y);
    int arg = 0;
    for (int o = 1; o < n_obj; ++o) {
        const float v = warped_prob(p, o, x, y);
        if (v > best) { best = v; arg = o; }
    }
    p.mask_out[(size_t)y * p.W + x] = best > p.seg_thr ? (unsigned char)(arg + 1) : 0;
}

int launch_paste_mask(const PasteParams &p, int B, void *stream) {
    if (B < 1 || B > CROP_MAX_B) return -1;
    dim3 grid((p.W + 255) / 256, p.H, B);
    hipLaunchKernelGGL(paste_mask_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_paste_labels(const PasteParams &p, int n_obj, void *stream) {
    if (n_obj < 1 || n_obj > CROP_MAX_B || !p.mask_out) return -1;
    dim3 grid((p.W + 255) / 256, p.H, 1);
    hipLaunchKernelGGL(paste_labels_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, n_obj);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
