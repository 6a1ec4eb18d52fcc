// l1_block.hip -- one identity-shortcut Bottleneck of layer1 (256 -> 64 -> 64 -> 256 planes at 63 x 63;
// experiments/siammask_sharp/resnet.py:80-103, blocks layer1.1 and layer1.2) as ONE launch, fp16.
//
// Why: at 63 x 63 x 256 channels the three convolutions of the block are HBM / launch-floor bound, not matrix bound
// (2.6 GFLOP per block and B = 8 frame batch, but 64 MB of activation traffic in three launches of 7-15 us: conv1 reads
// the 16 MB block input, conv3 reads it again as the residual and writes 16 MB; the 64-plane intermediates make two
// more round trips).  Here a workgroup owns an 8 x 16 patch of output positions of one image and keeps both 64-plane
// intermediates in LDS:
//   phase 1  t1 = ReLU(conv1(x) + b1) on the 10 x 18 halo of the patch (zero outside the image: conv2's padding pads
//            conv1's OUTPUT), K = 256 streamed in four 64-channel chunks (register prefetch of the next chunk)
//   phase 2  t2 = ReLU(conv2(t1) + b2): 3 x 3, the nine taps are row offsets into the t1 halo image in LDS
//   phase 3  out = ReLU(conv3(t2) + b3 + x): the residual rows come from L2 (phase 1 just read them), added in fp32
// MFMA v_mfma_f32_32x32x16_f16 throughout, operands in LDS with the same XOR swizzle as the conv kernels (16-byte slot
// ^ (row >> 1) & 7 on 128-byte rows).  Block input is read once (x 1.4 for the halo), the output written once.
// Rounding points are those of the per-layer path (fp16 after every fused conv+BN+ReLU), so both paths agree to fp32
// summation order.
#include <hip/hip_runtime.h>
#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int LB_TH = 8, LB_TW = 16;                 // output patch
constexpr int LB_HW = LB_TW + 2;                     // halo width 18
constexpr int LB_NH = (LB_TH + 2) * LB_HW;           // 180 halo positions
constexpr int LB_NHP = 192;                          // padded to six 32-row MFMA blocks
constexpr int LB_ROW = 128;                          // bytes per LDS row (64 halves)
constexpr int LB_T1 = LB_NHP * LB_ROW;               // 24 KB
constexpr int LB_XS = LB_NHP * LB_ROW;               // 24 KB per chunk buffer
constexpr int LB_WS = 64 * LB_ROW;                   // 8 KB per weight chunk buffer
constexpr int LB_T2 = 128 * LB_ROW;                  // 16 KB
constexpr int LB_W3 = 256 * LB_ROW;                  // 32 KB
constexpr int LB_OFF_T1 = 0, LB_OFF_XS = LB_T1, LB_OFF_WS = LB_OFF_XS + 2 * LB_XS, LB_OFF_T2 = LB_OFF_WS + 2 * LB_WS,
              LB_OFF_W3 = LB_OFF_T2 + LB_T2, LB_LDS = LB_OFF_W3 + LB_W3;
constexpr int LB_STG_LD = 132;                       // staging row stride (floats) of the 128-column epilogue halves
static_assert(128 * LB_STG_LD * 4 <= LB_OFF_WS, "the epilogue staging aliases t1 + both x chunk buffers");

__device__ __forceinline__ int lb_swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }
__device__ __forceinline__ half8 lb_frag(const unsigned char *base, int row, int ks, int fhalf) {
    return *(const half8 *)(base + row * LB_ROW + (lb_swz(row, ks * 2 + fhalf) << 4));
}
}  // namespace

__global__ __launch_bounds__(512, 1) void l1_block_kernel(const L1BlockParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LB_LDS];
    unsigned char *t1 = smem + LB_OFF_T1, *xs = smem + LB_OFF_XS, *ws = smem + LB_OFF_WS, *t2 = smem + LB_OFF_T2,
                  *w3s = smem + LB_OFF_W3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int tiles_x = (p.W + LB_TW - 1) / LB_TW, tiles_y = (p.H + LB_TH - 1) / LB_TH;
    int t = blockIdx.x;
    const int b = t / (tiles_x * tiles_y);
    t -= b * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * LB_TH, x0 = tx * LB_TW;
    const _Float16 *x = p.x + (size_t)b * p.H * p.W * 256;

    // ---------------- phase 1: t1 = ReLU(conv1(x) + b1) on the halo ----------------------------------------------
    // loads of one 64-channel chunk: 192 rows x 8 slots of x (3 vectors per thread) + 64 rows x 8 slots of w1 (1 per thread)
    uint4 xr[3], wr;
    const uint4 zero4 = {0u, 0u, 0u, 0u};
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int v = tid + i * 512, r = v >> 3, s = v & 7;
            const int hy = r / LB_HW, hx = r - hy * LB_HW;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = r < LB_NH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            xr[i] = ok ? *(const uint4 *)(x + ((size_t)iy * p.W + ix) * 256 + kc * 64 + s * 8) : zero4;
        }
        const int n = tid >> 3, s = tid & 7;
        wr = *(const uint4 *)(p.w1 + (size_t)n * p.kp1 + kc * 64 + s * 8);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int v = tid + i * 512, r = v >> 3, s = v & 7;
            *(uint4 *)(xs + buf * LB_XS + r * LB_ROW + (lb_swz(r, s) << 4)) = xr[i];
        }
        const int n = tid >> 3, s = tid & 7;
        *(uint4 *)(ws + buf * LB_WS + n * LB_ROW + (lb_swz(n, s) << 4)) = wr;
    };
    floatx16 acc1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < 4; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < 4) load_chunk(kc + 1);
        if (wave < 6) {
            const unsigned char *xa = xs + buf * LB_XS, *wb = ws + buf * LB_WS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 a = lb_frag(xa, wave * 32 + frow, ks, fhalf);
                const half8 b0 = lb_frag(wb, frow, ks, fhalf), b1 = lb_frag(wb, 32 + frow, ks, fhalf);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc1[1], 0, 0, 0);
            }
        }
        if (kc + 1 < 4) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // prefetch conv2's first tap and conv3's weights while t1 is written
    auto load_w2 = [&](int tap) {
        const int n = tid >> 3, s = tid & 7;
        wr = *(const uint4 *)(p.w2 + (size_t)n * p.kp2 + tap * 64 + s * 8);
    };
    auto store_w = [&](int buf) {
        const int n = tid >> 3, s = tid & 7;
        *(uint4 *)(ws + buf * LB_WS + n * LB_ROW + (lb_swz(n, s) << 4)) = wr;
    };
    load_w2(0);
    uint4 w3r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid + i * 512, n = v >> 3, s = v & 7;
        w3r[i] = *(const uint4 *)(p.w3 + (size_t)n * p.kp3 + s * 8);
    }
    if (wave < 6) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = j * 32 + frow;
            const float bias = p.b1[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                const int hy = row / LB_HW, hx = row - hy * LB_HW;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                const bool ok = row < LB_NH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const float v = ok ? fmaxf(acc1[j][r] + bias, 0.f) : 0.f;
                *(_Float16 *)(t1 + row * LB_ROW + (lb_swz(row, n >> 3) << 4) + (n & 7) * 2) = (_Float16)v;
            }
        }
    }
    store_w(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid + i * 512, n = v >> 3, s = v & 7;
        *(uint4 *)(w3s + n * LB_ROW + (lb_swz(n, s) << 4)) = w3r[i];
    }
    __syncthreads();

    // ---------------- phase 2: t2 = ReLU(conv2(t1) + b2), 3x3 pad 1 ---------------------------------------------
    floatx16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    {
        const int rb = wave >> 1, nb = wave & 1;
        const int pp = rb * 32 + frow, oy = pp >> 4, ox = pp & 15;          // this lane's A row: output position
        for (int tap = 0; tap < 9; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < 9) load_w2(tap + 1);
            const int ky = tap / 3, kx = tap - ky * 3;
            const int arow = (oy + ky) * LB_HW + ox + kx;
            const unsigned char *wb = ws + buf * LB_WS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 a = lb_frag(t1, arow, ks, fhalf);
                const half8 bb = lb_frag(wb, nb * 32 + frow, ks, fhalf);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc2, 0, 0, 0);
            }
            if (tap + 1 < 9) store_w(buf ^ 1);
            __syncthreads();
        }
        const int n = nb * 32 + frow;
        const float bias = p.b2[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            *(_Float16 *)(t2 + row * LB_ROW + (lb_swz(row, n >> 3) << 4) + (n & 7) * 2) = (_Float16)fmaxf(acc2[r] + bias, 0.f);
        }
    }
    __syncthreads();

    // ---------------- phase 3: out = ReLU(conv3(t2) + b3 + x) ---------------------------------------------------
    floatx16 acc3[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
    const int rb2 = wave >> 2, cb = wave & 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        half8 a[2], bb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = lb_frag(t2, rb2 * 64 + i * 32 + frow, ks, fhalf);
#pragma unroll
        for (int j = 0; j < 2; ++j) bb[j] = lb_frag(w3s, cb * 64 + j * 32 + frow, ks, fhalf);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc3[i][j], 0, 0, 0);
    }
    float *stg = (float *)smem;                          // [128 rows][LB_STG_LD]: aliases t1 + x chunk buffers (dead now)
    _Float16 *out = p.out + (size_t)b * p.H * p.W * 256;
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                 // staging free (h = 0: everybody is past phase 2's t1 reads)
        if ((cb >> 1) == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = (cb & 1) * 64 + j * 32 + frow;
                    const float bias = p.b3[h * 128 + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb2 * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                        stg[row * LB_STG_LD + col] = acc3[i][j][r] + bias;
                    }
                }
        }
        __syncthreads();
        // 128 rows x 16 groups of 8 channels: 4 items per thread, 16-byte residual load + 16-byte store each
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid + i * 512, row = v >> 4, g = v & 15;
            const int oy = y0 + (row >> 4), ox = x0 + (row & 15);
            if (oy < p.H && ox < p.W) {
                const size_t off = ((size_t)oy * p.W + ox) * 256 + h * 128 + g * 8;
                const half8 res = *(const half8 *)(x + off);
                const floatx4 s0 = *(const floatx4 *)(stg + row * LB_STG_LD + g * 8), s1 = *(const floatx4 *)(stg + row * LB_STG_LD + g * 8 + 4);
                half8 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = (_Float16)fmaxf(s0[q] + (float)res[q], 0.f);
                    o[q + 4] = (_Float16)fmaxf(s1[q] + (float)res[q + 4], 0.f);
                }
                *(half8 *)(out + off) = o;
            }
        }
    }
}

int launch_l1_block(const L1BlockParams &p, void *stream) {
    if (p.B < 1 || p.H < 1 || p.W < 1 || !p.x || !p.out || p.x == p.out) return -1;
    const int tiles = ((p.H + LB_TH - 1) / LB_TH) * ((p.W + LB_TW - 1) / LB_TW);
    hipLaunchKernelGGL(l1_block_kernel, dim3(tiles * p.B), dim3(512), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
