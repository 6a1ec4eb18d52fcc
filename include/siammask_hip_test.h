/*
 * siammask_hip_test.h -- the TEST and MEASUREMENT entry points of libsiammask_hip.so (MI355X / gfx950 only).
 *
 * Nothing in here is needed to drive the model: a maintainer who replaces the arithmetic of
 * experiments/siammask_sharp/custom.py binds siammask_hip.h only.  This header is what the parity tests (tests/), the
 * measurement scripts (tools/measure/) and bench.py's roofline leg bind in addition:
 *   smk_tune / smk_tune_get            process-wide A/B knobs (every default is a measured choice)
 *   smk_profile / smk_profile_dump     HIP events around every kernel launch, algorithmic flops / bytes per layer
 *   smk_debug_read                     read an internal activation back (parity of p0 .. search, zf, xs, corr ...)
 *   smk_op_*                           ONE kernel on caller tensors (unit parity against the oracle)
 *   smk_bench_conv, smk_host_*         per-geometry timing; host-only walks of the packing / planning logic (CPU tests)
 * Same conventions as siammask_hip.h (plain pointers and sizes, 0 / negative SMK_E* codes, smk_last_error()).
 */
#ifndef SIAMMASK_HIP_TEST_H
#define SIAMMASK_HIP_TEST_H

#include "siammask_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* PROCESS-WIDE tuning knobs for A/B measurements (affect subsequently launched / captured work).  They are NOT per context: two
 * contexts (two models, two dtypes) in one process share them, and a change does not invalidate graphs a context has already
 * captured.  Product code never calls smk_tune -- every default is the measured choice; tests and tools/measure/ put back what
 * they change (smk_tune_get).  A per-context copy was considered and left out on purpose: the knobs select between compiled
 * kernels and launch rules, the contexts hold state (weights, arena, graphs), and nothing on the product path sets one.
 *   "pipe_join" 0|1 (pipelined frame step, smk_set_pipeline: join with the previous frame's tail by a cross-queue event wait | by the
 *   in-stream gate kernel (default))   "pipe_eager" bit 0 / 1 (its front / tail as eager launches instead of graphs)
 *   "wreg96" 0|1 (conv_wreg_kernel: 96 x 256 tiles where 128 x 256 would leave a partial round of workgroups; default 1)
 *   "force_tile" 0 auto | 1 128x128 | 2 128x64 | 3 64x128 | 4 64x64 | 5 256x128   "kt" 0|128|256 (K-tile bytes)
 *   "stages" 0|2|3|4 (LDS ring depth)     "xcd_mode" 0|1|2 (tile -> XCD order)   "min_blocks_x16" (tile thresholds)
 *   "merge" 0|1 (independent convolutions share a launch)   "nt_store" 0|1 (streaming stores of the mask logits)
 *   "buf_lds" 0|1 (LDS-DMA through buffer resources; default 1)   "xc_ch" 64|32 (banded dw-xcorr: channels per workgroup; null)
 *   "l1_fused" 0|1 (fp16: every layer1 Bottleneck as one launch, l1_block_kernel: weights in registers, conv1 / conv2 outputs in LDS;
 *   default 1)   "stem_fused" 0|1 (fp16: frame -> conv1 + BN + ReLU -> p0 -> maxpool -> x1 as one launch, stem_pool_kernel; default 1)
 *   "xc_full" 0|1|2 (dw-xcorr: 0 = 5-row bands (default, fastest), 1 = 13-row bands (input read 1.14x instead of 1.8x, slower),
 *   2 = 5-row bands with batched loads (measurement))
 *   "prio" -1..3 (s_setprio of the consumer waves; measured null)   "mask_overlap" 0|1 (mask head on a graph side
 *   branch; measured slower)   "concurrency" 0|1 (fork/join between independent launches; measured slower; applies to
 *   contexts created afterwards)
 *   "halo" 0|1|64|128 (3x3 stride-1 convolutions through conv3x3_halo_kernel: off | per-shape choice (default, fp16) |
 *   force that workgroup height)   "halo_db" 0|1 (double-buffered patch for launches of <= one workgroup per CU)
 *   "chain" 0|1 (fp16: Refine's nine sequential convolutions as one launch,
 *   refine_chain_kernel; default 1)   "ksplit" 0|1|2|4 (split-K across workgroups with a last-arrival reduction:
 *   off (default; measured a net loss at B=8) | auto for long-K few-tile launches | forced factor).
 *   "a_stage" 0|1 (conv_wreg_kernel / conv_seq_kernel producers: activation rows by LDS-DMA with the swizzle on the source
 *   address | global -> VGPR in ascending lane order, swizzle applied by ds_write_b128; same LDS image, bit-identical results).
 *   "seq_fuse" 0..3 (conv_seq_kernel: a Bottleneck's conv3 + the 1x1 convolution that reads it -- the next block's conv1, adjust -- as
 *   ONE tile routine on 32-row tiles, c3c1_tile.inc: off | every pair the routine has a shape for (default) | layer3's pairs only |
 *   same as 1);   "seq_ds128" 0|1 (layer2.0's 3x3 stride-2 shortcut on 128x128 sequence tiles)
 *   smk_tune_get("seq_fused_last") = pairs fused in the sequence launched last (read-only diagnostic).
 * Environment: SMK_CHAIN_CLK=1 makes eager (non-graph) runs print the time workgroup 0 spends in each layer of
 * refine_chain_kernel to stderr (measurement aid). */
int smk_tune(const char *key, int value);
/* current value of a knob (tests and A/B scripts restore what they changed; also how a caller reads the defaults) */
int smk_tune_get(const char *key, int *value);

/* per-launch profiling: with enable != 0 every kernel launch is bracketed by HIP events on the
 * stream it is launched on (graph replay is bypassed while profiling).  smk_profile_dump
 * synchronises the device and writes a JSON array, one object per layer id in launch order:
 *   {"id","kernel","calls","ms" (sum of event durations),"flop","bytes"} where flop/bytes are
 * the ALGORITHMIC work of those launches (2*M*N*K; tensors read/written once), then resets.
 * enable = 1: per-LAYER attribution (a launch that merges several independent convolutions is split into its members);
 * enable = 2: per-LAUNCH attribution -- the launch structure of the timed path is kept (one record per merged launch, its
 * id the members' ids joined by '+'): what bench.py's per-kernel roofline table is made of. */
int smk_profile(smk_ctx *ctx, int enable);
int smk_profile_dump(smk_ctx *ctx, char *json_buf, int capacity);

/* read back an internal activation as f32 NCHW into a device buffer (parity tests only).
 * names: "p0","p1","p2","p3","search","zf","zk","xs","corr","head0"; batch = last batch.
 * *numel_out receives C*H*W per item; dst may be NULL to query the shape (c,h,w). */
/* test aid: raise the persistent sequence kernel's failure flag (device + host-mapped copy) exactly as the kernel does
 * (code 1 = uneven XCD placement, 2 = team-barrier time-out): the next sequence launch finds it set and returns at once, so
 * the failure path (smk_seq_sync_check, the fall-back to the per-layer kernels, siammask_amd.custom's transparent re-run)
 * can be exercised without provoking a real time-out. */
int smk_debug_seq_inject(smk_ctx *ctx, int code);
int smk_debug_read(smk_ctx *ctx, const char *name, float *dst_dev, int *c, int *h, int *w,
                   void *stream);

/* ---- per-op entry points (unit parity against the oracle; not used by the tools) --------
 * smk_op_conv2d: y = act(conv2d(x, w) + b [+ res]) on f32 NCHW device tensors, computed by
 * the same implicit-GEMM MFMA kernel the network uses (algo 0) or by the naive one-thread-
 * per-output kernel (algo 1).  w: host [Cout,Cin,k,k], b: host [Cout] or NULL,
 * res: device [B,Cout,Ho,Wo] or NULL (added before the ReLU).
 * smk_op_dw_xcorr <- models/rpn.py:32-38 conv2d_dw_group: x [B,C,H,W], k [B,C,kh,kw]. */
/* geometry of one convolution for the per-op entry points; tensors are f32 NCHW at this boundary */
typedef struct smk_conv_geom {
    int B, Cin, H, W;          /* input tensor [B,Cin,H,W]                                        */
    int Cout, k, stride, pad, dil;
    int relu;                  /* apply ReLU in the epilogue                                      */
    int res_mode;              /* 0 none, 1 add residual before ReLU, 2 add after ReLU            */
    int win;                   /* 1: the conv sees the Hl x Wl window of the input whose origin is
                                  (org_y + pos_y*pos_mul + pos_add, org_x + ...); outside the
                                  window AND outside the tensor reads as zero (F.pad + slice,
                                  custom.py:133-135; centre crop custom.py:21-24)                 */
    int ups;                   /* 1: the conv sees the input nearest-upsampled to Hl x Wl
                                  (F.upsample, custom.py:150-152)                                 */
    int Hl, Wl, org_y, org_x, pos_mul, pos_add;
    int cin_off, cin_len;      /* use channels [cin_off, cin_off+cin_len) (cin_len 0 = all)       */
} smk_conv_geom;

/* algo: low byte 0 = MFMA kernel, NHWC epilogue; 1 = naive kernel, NHWC epilogue;
 *                2 = MFMA kernel, NCHW-f32 epilogue; 3 = naive kernel, NCHW-f32 epilogue;
 *                4 = conv3x3_halo_kernel (tile code 1 = 128 rows, else 64); 5 = conv_wreg_kernel (tile code 1..8 =
 *                64x256 64x128 64x64 128x256 128x128 128x64 96x256 32x64; ring-depth bits: 0 the library's choice, 1 eight k-steps ahead on every shape (measurement builds; else = 2), 2 / 3 = a three- / four-deep ring); 6 = conv_pp_kernel
 *                (256 x 256 tiles, fp16 only, conv_pp.hip);
 *       second byte: bits 0-3 tile override 0 auto, 1 128x128, 2 128x64, 3 64x128, 4 64x64, 5 256x128;
 *                    bits 4-5 K tile 0 auto, 1 = 128 B, 2 = 256 B; bits 6-7 LDS ring depth
 *                    0 auto, 1..3 = 2..4 stages.
 * w_host [Cout,cin_len,k,k], b_host [Cout] or NULL (host); x_dev, res_dev [B,Cout,Ho,Wo],
 * y_dev (device); pos_host: B (y,x) pairs or NULL.  Synchronises the stream (test helper). */
int smk_op_conv2d_ex(int dtype, int algo, const smk_conv_geom *g, const float *x_dev,
                     const float *w_host, const float *b_host, const float *res_dev,
                     const int32_t *pos_host, float *y_dev, void *stream);
int smk_op_conv2d(int dtype, int algo, const float *x_dev, int B, int Cin, int H, int W,
                  const float *w_host, const float *b_host, int Cout, int k, int stride,
                  int pad, int dil, int relu, const float *res_dev, float *y_dev,
                  void *stream);
/* smk_op_conv_seq: n <= 36 convolutions as ONE persistent conv_seq_kernel launch (fp16; the kernel that runs ResNet layer2 /
 * layer3 / adjust at B = 8, experiments/siammask_sharp/resnet.py:64-103,159-165) -- unit parity of every tile configuration,
 * of the residual path, of independent members (no barrier between them) and of the split team barrier.
 *   layer i reads the sequence input x (src = -1) or the output of layer src < i; g is the geometry of ITS input (g.B the same
 *   for all layers; no windows / upsampling); g.res_mode != 0 adds the tensor res_src (-1 = x, j < i = output of layer j,
 *   same shape as the output) before / after the ReLU; sync != 0 puts a team barrier behind the layer (needed whenever a later
 *   layer reads what this one or an earlier unsynchronised one wrote); cfg = tile code 0 64x256, 1 64x128, 2 64x64,
 *   3 128x256, 4 128x128, 5..8 = measurement variants of 64x128 (5 deeper rings; 6 / 7 / 8 without activation refills /
 *   weight refills / MFMA: wrong results by construction), -1 = the engine's choice; kstag = K-loop stagger
 *   1 / 0, -1 = the engine's choice.  w_host [Cout,Cin,k,k], b_host [Cout] or NULL; y_dev: device f32 NCHW output or NULL.
 * The launch is repeated `iters` times; *usec_out (optional) = average microseconds of launches 2..iters; clk_us_out
 * (optional, [2*n]) = per layer, the time (team 0, slot 0) spent in its tiles and in the barrier arrival, of the last launch.
 * Like the engine's own lists, the list goes through the pair fusion (smk_tune "seq_fuse": a Bottleneck's conv3 + the 1x1
 * convolution that reads it run as one tile routine, c3c1_tile.inc; cfg = -1 on both); *n_fused_out (optional) = pairs fused.
 * Synchronises the stream (test helper). */
typedef struct smk_seq_op {
    smk_conv_geom g;
    int src, res_src, sync, cfg, kstag;
    const float *w_host, *b_host;
    float *y_dev;
} smk_seq_op;
int smk_op_conv_seq(const smk_seq_op *ops, int n, const float *x_dev, int iters, float *usec_out,
                    float *clk_us_out, int *n_fused_out, void *stream);
int smk_op_dw_xcorr(int dtype, const float *x_dev, const float *k_dev, int B, int C,
                    int H, int W, int kh, int kw, float *y_dev, void *stream);
int smk_op_maxpool3x3s2(int dtype, const float *x_dev, int B, int C, int H, int W,
                        float *y_dev, void *stream);

/* measurement aid: time `iters` back-to-back launches of the MFMA conv kernel for geometry g
 * (random f16/f32 operands allocated internally, NHWC epilogue unless algo low byte is 2) with
 * HIP events on `stream`; *usec_out = average microseconds per launch.  algo as above. */
int smk_bench_conv(int dtype, int algo, const smk_conv_geom *g, int with_res, int iters,
                   float *usec_out, void *stream);

/* host-only (no GPU): y = epilogue(conv(x, w) + b) computed on the HOST by walking the packed
 * weight matrix with the device kernels' own row/tap decode + gather-offset functions.
 * Lets the CPU test-suite verify packing order, padding, windows and upsampling.
 * All pointers are host memory. */
int smk_host_conv2d_ex(const smk_conv_geom *g, const float *x, const float *w, const float *b,
                       const float *res, const int32_t *pos, float *y);

/* Host only: which kernel the engine picks for ONE convolution of this geometry and batch (g->B streams) under the current
 * smk_tune knobs -- *kernel = 0 conv_igemm_kernel, 1 conv3x3_halo_kernel, 2 conv_wreg_kernel, 3 conv_pp_kernel; *bm x *bn = workgroup shape;
 * *seq_cfg = tile code inside a persistent per-XCD sequence (0 64x256, 1 64x128, 2 64x64, 3 128x256, 4 128x128) or -1 when
 * the layer cannot be part of one.  Lets the CPU test-suite pin the measured layer rules (profiles/r02_producer_waves_*). */
int smk_host_plan_conv(const smk_conv_geom *g, int dtype, int with_res, int *kernel, int *bm, int *bn, int *seq_cfg);

#ifdef __cplusplus
}
#endif
#endif /* SIAMMASK_HIP_TEST_H */
