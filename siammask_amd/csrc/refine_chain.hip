// refine_chain.hip -- the sequential tail of Refine (experiments/siammask_sharp/custom.py:150-153) as ONE
// launch, one workgroup per stream, every intermediate activation resident in LDS (fp16 path).
//
//   out = post0(up31(h2(deconv_out) + V2))      h2 = conv3x3+ReLU, conv3x3+ReLU   (32 ch @ 15x15)
//   out = post1(up61(h1(out)        + V1))      h1 = ...                          (16 ch @ 31x31)
//   out = post2(up127(h0(out)       + V0))      h0 = ...                          ( 4 ch @ 61x61)
//
// V2 / V1 / V0 (= v2 / v1 / v0 of the reference applied to the feature windows) do not depend on this
// chain; the engine computes them beforehand in merged launches of the generic kernel and this kernel
// adds them after h*.2's ReLU.  Nine dependent 3x3 convolutions on 225 .. 16129 pixels with 32 .. 1
// output channels are ~16 MMAC per stream: as nine launches they cost nine launch floors (~8 us each on
// MI355X, nothing to do with the arithmetic); here they are nine LDS-to-LDS passes.
//
// One pass = implicit GEMM  [pixels x 9*Cin] x [9*Cin x Cout]  on v_mfma_f32_16x16x32_f16:
//   A fragment: lane -> pixel (lane & 15), eight consecutive K elements (lane >> 4) -- one 16-byte LDS read
//               of eight channels of one tap;
//   B fragment: lane -> output channel (lane & 15), same K slice, from the layer's weights staged in LDS;
//   C: col = lane & 15 (channel), row = 4*(lane >> 4) + reg (pixel).
// Activations live in LDS as zero-bordered [(H+2) x (H+2) x C] fp16 images so the 3x3 taps of an ordinary
// layer are nine constant offsets; the layers that read through the nearest-neighbour upsampling
// (sy = iy*Hs/H, like the generic kernel) bounds-check per tap instead.  Every layer output is rounded to
// fp16 exactly where the per-layer path stores fp16 activations, so the two paths share rounding points.
#include <hip/hip_runtime.h>

#include "smk_kernels.h"

namespace smk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RC_NT = 1024;                          // threads per workgroup (16 waves)
constexpr int RC_BUF = 33 * 33 * 16 * 2;             // largest bordered activation image (bytes)
constexpr int RC_WROWS = 32, RC_WPITCH = 288 + 8;    // weight stage: [<=32 rows][<=288 K + pad] halfs ...
constexpr int RC_WSTAGE = RC_WROWS * RC_WPITCH * 2 + RC_WROWS * 4;     // ... + 32 fp32 biases
constexpr int RC_TAB = 132;                          // upsampling index tables: rows, columns (ints)
constexpr int RC_LDS = 3 * RC_BUF + 2 * RC_WSTAGE + 2 * RC_TAB * 4;    // three images + double-buffered weights + tables

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for
// the global fetches this kernel deliberately keeps in flight across layers (measured: ~2.5 us per layer).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int CIN, int COUT>
struct WGeo {
    static constexpr int K = 9 * CIN, KSTEPS = (K + 31) / 32, KP = KSTEPS * 32 + 8, NT = (COUT + 15) / 16;
    static constexpr int G = KSTEPS * 8;             // 4-half groups per staged weight row
    static constexpr int TOTAL = NT * 16 * G, PER = (TOTAL + RC_NT - 1) / RC_NT;
    static_assert(CIN == 32 || CIN == 16 || CIN == 4, "input channels");
    static_assert(NT * 16 <= RC_WROWS && KP <= RC_WPITCH, "weight stage too small");
};

// a layer's weights + bias travel global -> registers (issued a whole layer ahead) -> LDS stage
template <int CIN, int COUT>
struct WRegs {
    half4 v[WGeo<CIN, COUT>::PER];
    float bias;
};
template <int CIN, int COUT>
__device__ __forceinline__ void w_load(const RefineChainLayer &L, WRegs<CIN, COUT> &r) {
    typedef WGeo<CIN, COUT> WG;
    const _Float16 *w = (const _Float16 *)L.w;
#pragma unroll
    for (int j = 0; j < WG::PER; ++j) {
        const int i = threadIdx.x + j * RC_NT;
        const int n = i / WG::G, g = i - n * WG::G;
        const int k = g * 4, tap = k / CIN, c = k - tap * CIN;
        r.v[j] = half4{0, 0, 0, 0};
        if (i < WG::TOTAL && n < COUT && tap < 9) r.v[j] = *(const half4 *)(w + (size_t)n * L.Kpad + tap * L.Ci + c);
    }
    r.bias = (int)threadIdx.x < COUT ? L.bias[threadIdx.x] : 0.f;
}
template <int CIN, int COUT>
__device__ __forceinline__ void w_store(_Float16 *wl, const WRegs<CIN, COUT> &r) {
    typedef WGeo<CIN, COUT> WG;
#pragma unroll
    for (int j = 0; j < WG::PER; ++j) {
        const int i = threadIdx.x + j * RC_NT;
        const int n = i / WG::G, g = i - n * WG::G;
        if (i < WG::TOTAL) *(half4 *)(wl + n * WG::KP + g * 4) = r.v[j];
    }
    if (threadIdx.x < RC_WROWS) ((float *)(wl + RC_WROWS * RC_WPITCH))[threadIdx.x] = r.bias;
}

template <int C, int O>
using WR = WRegs<(C ? C : 4), (C ? O : 1)>;          // C == 0: "no layer" placeholder

// V_x = ReLU(v_x.2(...)) [VH*VH pixels][max(VC,8) channels] global -> registers -> the bordered LDS image that
// h_x.2 accumulates into
template <int VH, int VC>
struct VRegs {
    static constexpr int VPP = (VC < 8 ? 8 : VC) / 8, TOTAL = VH * VH * VPP, PER = (TOTAL + RC_NT - 1) / RC_NT;
    half8 v[PER > 0 ? PER : 1];
};
template <int VH, int VC>
__device__ __forceinline__ void v_load(const _Float16 *src, VRegs<VH, VC> &r) {
#pragma unroll
    for (int j = 0; j < VRegs<VH, VC>::PER; ++j) {
        const int i = threadIdx.x + j * RC_NT;
        r.v[j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (i < VRegs<VH, VC>::TOTAL) r.v[j] = *(const half8 *)(src + (size_t)i * 8);
    }
}
template <int VH, int VC>
__device__ __forceinline__ void v_store(_Float16 *img, const VRegs<VH, VC> &r) {
    constexpr int VPP = VRegs<VH, VC>::VPP;
#pragma unroll
    for (int j = 0; j < VRegs<VH, VC>::PER; ++j) {
        const int i = threadIdx.x + j * RC_NT;
        if (i < VRegs<VH, VC>::TOTAL) {
            const int px = i / VPP, part = i - px * VPP;
            const int y = px / VH, x = px - y * VH;
            _Float16 *d = img + ((y + 1) * (VH + 2) + x + 1) * VC + part * 8;
            if (VC >= 8) *(half8 *)d = r.v[j];
            else *(half4 *)d = half4{r.v[j][0], r.v[j][1], r.v[j][2], r.v[j][3]};
        }
    }
}

// nearest upsampling SRC -> H of a bordered source image as two tables indexed by (output coordinate + tap):
// element offset of the source row / column, or of the zero border when the tap falls outside the H x H image
template <int H, int SRC, int C>
__device__ __forceinline__ void build_up_tables(int *tab) {
    static_assert(H + 2 <= RC_TAB, "table too small");
    const int i = threadIdx.x;
    if (i < H + 2) {
        const int iy = i - 1;
        const int sv = (unsigned)iy < (unsigned)H ? (iy * SRC) / H + 1 : 0;
        tab[i] = sv * (SRC + 2) * C;
        tab[RC_TAB + i] = sv * C;
    }
}

template <int H, int C>
__device__ __forceinline__ void zero_image(_Float16 *img) {
    constexpr int N16 = ((H + 2) * (H + 2) * C * 2 + 15) / 16;
    const floatx4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < N16; i += RC_NT) ((floatx4 *)img)[i] = z;
}

// One 3x3 (pad 1) layer:  in (LDS) -> out (LDS image or global fp32 plane)
//   CIN / COUT  real channels (CIN = LDS channel count of the input image)
//   H           output height = width;  SRC > 0: the input image is SRC x SRC and is read through
//               nearest upsampling to H x H, SRC == 0: the input image is H x H
//   RES         `out` already holds the other branch (V_x): add to it after the ReLU
//   NCIN/NCOUT  next layer's weights: already in flight to the registers `wnext`, staged into LDS after this
//               layer's arithmetic
//   FCIN/FCOUT  the layer after that: its fetch to `wfar` starts before this layer's arithmetic (a layer can
//               be shorter than one global-memory round trip, two never are)
//   VH/VC       V_x for the layer after this one: global -> registers -> the image `vimg` likewise
template <int CIN, int COUT, int H, int SRC, bool RELU, bool RES, bool OUT_GLOBAL, int NCIN, int NCOUT, int FCIN,
          int FCOUT, int VH, int VC>
__device__ __forceinline__ void chain_layer(const _Float16 *in, _Float16 *out, const _Float16 *wl, _Float16 *wl_next,
                                            const WR<NCIN, NCOUT> &wnext, const RefineChainLayer &Lfar,
                                            WR<FCIN, FCOUT> &wfar, const _Float16 *vsrc, _Float16 *vimg,
                                            const int *tab, float *gout) {
    typedef WGeo<CIN, COUT> WG;
    constexpr int KSTEPS = WG::KSTEPS, KP = WG::KP, NT = WG::NT, M = H * H, MT = (M + 15) / 16;
    constexpr int HI = SRC > 0 ? SRC : H, W2 = HI + 2;       // input image geometry (bordered width)
    constexpr int WO = H + 2;
    constexpr bool PRELOAD = KSTEPS * NT <= 10;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- prologue: clear the output image (its border is the next layer's zero padding); start the fetches
    VRegs<VH ? VH : 1, VH ? VC : 8> vn;
    if (FCIN) w_load<FCIN ? FCIN : 4, FCIN ? FCOUT : 1>(Lfar, wfar);
    if (VH) v_load<VH ? VH : 1, VH ? VC : 8>(vsrc, vn);
    if (!OUT_GLOBAL && !RES) zero_image<H, COUT>(out);
    if (VH) zero_image<VH ? VH : 1, VH ? VC : 8>(vimg);
    if (SRC > 0) build_up_tables<H, (SRC > 0 ? SRC : 1), CIN>((int *)tab);
    lds_barrier();

    // ---- per-lane constants
    const int fr = lane & 15, kq = lane >> 4;
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = ((const float *)(wl + RC_WROWS * RC_WPITCH))[nt * 16 + fr];
    // K slice of this lane at step ks starts at k0 = ks*32 + kq*8: its tap and first channel as loop-invariant
    // offsets.  In the K padding (tap > 8) the weights are zero, any finite activation will do: clamp the tap.
    static_assert(CIN >= 8, "the 4-channel layers go through chain_layer_c4");
    int kh_[KSTEPS], kw_[KSTEPS], koff[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int k0 = ks * 32 + kq * 8;
        int tap = k0 / CIN;
        const int c0 = k0 - tap * CIN;
        tap = tap > 8 ? 8 : tap;
        kh_[ks] = (tap * 11) >> 5;
        kw_[ks] = tap - 3 * kh_[ks];
        koff[ks] = SRC > 0 ? c0 : (kh_[ks] * W2 + kw_[ks]) * CIN + c0;
    }
    half8 wf[PRELOAD ? KSTEPS : 1][NT];
    if (PRELOAD) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[ks][nt] = *(const half8 *)(wl + (nt * 16 + fr) * KP + ks * 32 + kq * 8);
    }

    for (int mt = wave; mt < MT; mt += RC_NT / 64) {
        int m = mt * 16 + fr;
        m = m < M ? m : M - 1;
        const int oy = m / H, ox = m - oy * H;
        // element offset (halfs) of this lane's K slice in the bordered input image
        const int pbase = (oy * W2 + ox) * CIN;
        auto a_off = [&](int ks) {
            if (SRC > 0) return tab[oy + kh_[ks]] + tab[RC_TAB + ox + kw_[ks]] + koff[ks];
            return pbase + koff[ks];
        };
        floatx4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const half8 a = *(const half8 *)(in + a_off(ks));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8 b = PRELOAD ? wf[PRELOAD ? ks : 0][nt]
                                        : *(const half8 *)(wl + (nt * 16 + fr) * KP + ks * 32 + kq * 8);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[nt], 0, 0, 0);
            }
        }
        // ---- epilogue: C[row = pixel 4*kq + i][col = channel fr]
        const int r0 = mt * 16 + 4 * kq;
        const int y0 = r0 / H, x0 = r0 - y0 * H;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + fr;
            if (OUT_GLOBAL) {
                if (n == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r0 + i < M) gout[r0 + i] = acc[nt][i] + bias[nt];
                }
            } else if (n < COUT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (r0 + i < M) {
                        const int wrap = x0 + i >= H;                     // H > 4: at most one row wrap
                        _Float16 *o = out + ((y0 + wrap + 1) * WO + x0 + i - (wrap ? H : 0) + 1) * COUT + n;
                        float v = acc[nt][i] + bias[nt];
                        if (RELU) v = fmaxf(v, 0.f);
                        if (RES) v += (float)*o;
                        *o = (_Float16)v;
                    }
                }
            }
        }
    }
    // ---- tail: stage the next layer's weights (fetched a layer ago) and V_x
    if (NCIN) w_store<NCIN ? NCIN : 4, NCIN ? NCOUT : 1>(wl_next, wnext);
    if (VH) v_store<VH ? VH : 1, VH ? VC : 8>(vimg, vn);
    lds_barrier();
}

// The 4-input-channel layers (h0.0, h0.2, post2) on the vector ALU: with Cin = 4 and Cout <= 4 an MFMA pass
// spends ~10x more instructions on im2col addressing than on arithmetic (measured: the whole chain was
// instruction-issue bound at 102 us, post2 alone 60%), so here one lane owns one output pixel: nine 8-byte
// LDS reads (constant offsets from one base address), 18*COUT v_dot2_f32_f16 against weights held in VGPRs.
template <int COUT, int H, int SRC, bool RELU, bool RES, bool OUT_GLOBAL, int NCIN, int NCOUT, int FCIN, int FCOUT,
          int VH, int VC>
__device__ __forceinline__ void chain_layer_c4(const _Float16 *in, _Float16 *out, const _Float16 *wl, _Float16 *wl_next,
                                               const WR<NCIN, NCOUT> &wnext, const RefineChainLayer &Lfar,
                                               WR<FCIN, FCOUT> &wfar, const _Float16 *vsrc, _Float16 *vimg,
                                               const int *tab, float *gout) {
    typedef WGeo<4, COUT> WG;
    constexpr int KP = WG::KP, M = H * H;
    constexpr int HI = SRC > 0 ? SRC : H, W2 = HI + 2, WO = H + 2;
    static_assert(COUT == 4 || COUT == 1, "output channels");
    const int tid = threadIdx.x;

    VRegs<VH ? VH : 1, VH ? VC : 8> vn;
    if (FCIN) w_load<FCIN ? FCIN : 4, FCIN ? FCOUT : 1>(Lfar, wfar);
    if (VH) v_load<VH ? VH : 1, VH ? VC : 8>(vsrc, vn);
    if (!OUT_GLOBAL && !RES) zero_image<H, COUT>(out);
    if (VH) zero_image<VH ? VH : 1, VH ? VC : 8>(vimg);
    if (SRC > 0) build_up_tables<H, (SRC > 0 ? SRC : 1), 4>((int *)tab);
    lds_barrier();

    half2v wv[COUT][9][2];                                     // [out channel][tap][channel pair]
    float bias[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        bias[co] = ((const float *)(wl + RC_WROWS * RC_WPITCH))[co];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const half4 w4 = *(const half4 *)(wl + co * KP + t * 4);
            wv[co][t][0] = half2v{w4[0], w4[1]};
            wv[co][t][1] = half2v{w4[2], w4[3]};
        }
    }
    for (int px = tid; px < M; px += RC_NT) {
        const int oy = px / H, ox = px - oy * H;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        if (SRC > 0) {
            int ro[3], cl[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ro[t] = tab[oy + t];
                cl[t] = tab[RC_TAB + ox + t];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const half4 a = *(const half4 *)(in + ro[t / 3] + cl[t % 3]);
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    acc[co] = __builtin_amdgcn_fdot2(half2v{a[0], a[1]}, wv[co][t][0], acc[co], false);
                    acc[co] = __builtin_amdgcn_fdot2(half2v{a[2], a[3]}, wv[co][t][1], acc[co], false);
                }
            }
        } else {
            const _Float16 *base = in + (oy * W2 + ox) * 4;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const half4 a = *(const half4 *)(base + ((t / 3) * W2 + t % 3) * 4);
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    acc[co] = __builtin_amdgcn_fdot2(half2v{a[0], a[1]}, wv[co][t][0], acc[co], false);
                    acc[co] = __builtin_amdgcn_fdot2(half2v{a[2], a[3]}, wv[co][t][1], acc[co], false);
                }
            }
        }
        if (OUT_GLOBAL) {
            gout[px] = acc[0] + bias[0];
        } else {
            _Float16 *o = out + ((oy + 1) * WO + ox + 1) * COUT;
            float v[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                v[co] = acc[co] + bias[co];
                if (RELU) v[co] = fmaxf(v[co], 0.f);
            }
            if (COUT == 4) {
                if (RES) {
                    const half4 r = *(const half4 *)o;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) v[co] += (float)r[co];
                }
                *(half4 *)o = half4{(_Float16)v[0], (_Float16)v[COUT > 1 ? 1 : 0], (_Float16)v[COUT > 2 ? 2 : 0],
                                    (_Float16)v[COUT > 3 ? 3 : 0]};
            } else {
                *o = (_Float16)v[0];
            }
        }
    }
    if (NCIN) w_store<NCIN ? NCIN : 4, NCIN ? NCOUT : 1>(wl_next, wnext);
    if (VH) v_store<VH ? VH : 1, VH ? VC : 8>(vimg, vn);
    lds_barrier();
}

}  // namespace

// TIMED: workgroup 0 also records a timestamp per layer (SMK_CHAIN_CLK=1; see engine.cpp)
template <bool TIMED>
__global__ __launch_bounds__(1024) void refine_chain_kernel(const RefineChainParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RC_LDS];
    _Float16 *bA = (_Float16 *)smem, *bB = (_Float16 *)(smem + RC_BUF), *bC = (_Float16 *)(smem + 2 * RC_BUF);
    _Float16 *w0 = (_Float16 *)(smem + 3 * RC_BUF), *w1 = (_Float16 *)(smem + 3 * RC_BUF + RC_WSTAGE);
    const int b = blockIdx.x;
    const _Float16 *v2 = (const _Float16 *)p.v2 + (size_t)b * (225 * 32);
    const _Float16 *v1 = (const _Float16 *)p.v1 + (size_t)b * (961 * 16);
    const _Float16 *v0 = (const _Float16 *)p.v0 + (size_t)b * (3721 * 8);
    float *gout = p.out + (size_t)b * (127 * 127);
    int *tab = (int *)(smem + 3 * RC_BUF + 2 * RC_WSTAGE);
    unsigned long long *clk = TIMED && p.clk && blockIdx.x == 0 && threadIdx.x == 0 ? p.clk : nullptr;
    if (TIMED && clk) clk[0] = wall_clock64();
    // weights of layer i+1 / i+2 in flight (registers) while layer i computes
    WRegs<32, 32> r1;
    WRegs<32, 16> r2;
    WRegs<16, 16> r3, r4;
    WRegs<16, 4> r5;
    WRegs<4, 4> r6, r7;
    WRegs<4, 1> r8, none;
    // deconv output [15*15][32] -> bordered image A; first layer's weights
    {
        WRegs<32, 32> r0;
        VRegs<15, 32> dr;
        w_load<32, 32>(p.L[0], r0);
        v_load<15, 32>((const _Float16 *)p.d + (size_t)b * 7200, dr);
        w_load<32, 32>(p.L[1], r1);
        zero_image<15, 32>(bA);
        lds_barrier();
        w_store<32, 32>(w0, r0);
        v_store<15, 32>(bA, dr);
        // chain_layer's prologue barrier orders these writes before the first reads
    }
    if (TIMED && clk) clk[1] = wall_clock64();
    //          CIN COUT  H  SRC  RELU   RES  GLOBAL  next W  W after  next V
    chain_layer<32, 32, 15, 0, true, false, false, 32, 32, 32, 16, 15, 32>(bA, bB, w0, w1, r1, p.L[2], r2, v2, bC, tab, nullptr);  // h2.0
    if (TIMED && clk) clk[2] = wall_clock64();
    chain_layer<32, 32, 15, 0, true, true, false, 32, 16, 16, 16, 0, 0>(bB, bC, w1, w0, r2, p.L[3], r3, nullptr, nullptr, tab, nullptr);  // h2.2 + V2
    if (TIMED && clk) clk[3] = wall_clock64();
    chain_layer<32, 16, 31, 15, false, false, false, 16, 16, 16, 16, 0, 0>(bC, bA, w0, w1, r3, p.L[4], r4, nullptr, nullptr, tab, nullptr);  // post0(up31)
    if (TIMED && clk) clk[4] = wall_clock64();
    chain_layer<16, 16, 31, 0, true, false, false, 16, 16, 16, 4, 31, 16>(bA, bB, w1, w0, r4, p.L[5], r5, v1, bC, tab, nullptr);  // h1.0
    if (TIMED && clk) clk[5] = wall_clock64();
    chain_layer<16, 16, 31, 0, true, true, false, 16, 4, 4, 4, 0, 0>(bB, bC, w0, w1, r5, p.L[6], r6, nullptr, nullptr, tab, nullptr);  // h1.2 + V1
    if (TIMED && clk) clk[6] = wall_clock64();
    chain_layer<16, 4, 61, 31, false, false, false, 4, 4, 4, 4, 0, 0>(bC, bA, w1, w0, r6, p.L[7], r7, nullptr, nullptr, tab, nullptr);  // post1(up61)
    if (TIMED && clk) clk[7] = wall_clock64();
    chain_layer_c4<4, 61, 0, true, false, false, 4, 4, 4, 1, 61, 4>(bA, bB, w0, w1, r7, p.L[8], r8, v0, bC, tab, nullptr);  // h0.0
    if (TIMED && clk) clk[8] = wall_clock64();
    chain_layer_c4<4, 61, 0, true, true, false, 4, 1, 0, 0, 0, 0>(bB, bC, w1, w0, r8, p.L[8], none, nullptr, nullptr, tab, nullptr);  // h0.2 + V0
    if (TIMED && clk) clk[9] = wall_clock64();
    chain_layer_c4<1, 127, 61, false, false, true, 0, 0, 0, 0, 0, 0>(bC, nullptr, w0, nullptr, none, p.L[8], none, nullptr, nullptr, tab, gout);  // post2
    if (TIMED && clk) clk[10] = wall_clock64();
}

int launch_refine_chain(const RefineChainParams &p, void *stream) {
    if (p.v2_cs != 32 || p.v1_cs != 16 || p.v0_cs != 8) return -1;      // the layouts v_load assumes
    if (p.clk) hipLaunchKernelGGL(refine_chain_kernel<true>, dim3(p.B), dim3(RC_NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(refine_chain_kernel<false>, dim3(p.B), dim3(RC_NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
