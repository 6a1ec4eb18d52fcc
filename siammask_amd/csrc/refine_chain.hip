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

#include "refine_chain_body.inc"

template <bool TIMED>
__global__ __launch_bounds__(1024) void refine_chain_kernel(const RefineChainParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RC_LDS];
    refine_chain_body<TIMED>(p, (int)blockIdx.x, smem);
}

int launch_refine_chain(const RefineChainParams &p, void *stream) {
    if (p.v2_cs != 32 || p.v1_cs != 16 || p.v0_cs != 8) return -1;      // the layouts v_load assumes
    if (p.clk) hipLaunchKernelGGL(refine_chain_kernel<true>, dim3(p.B), dim3(RC_NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(refine_chain_kernel<false>, dim3(p.B), dim3(RC_NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace smk
