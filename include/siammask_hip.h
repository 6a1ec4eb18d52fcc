/*
 * siammask_hip.h -- C ABI of libsiammask_hip.so (MI355X / gfx950 only).
 *
 * The reference (foolwood/SiamMask) has no native layer on its inference path: the path's
 * arithmetic is dispatched from Python into PyTorch (torch==0.4.1, requirements.txt:6).
 * This library replaces everything *below* the reference's drop-in boundary
 *     experiments/siammask_sharp/custom.py:173-190   Custom.template / track / track_mask / track_refine
 *     experiments/siammask_base/custom.py:100-112    (3-branch variant, 63x63 mask head)
 *     experiments/siamrpn_resnet/custom.py:87-93     (box-only variant)
 * with hand-written HIP kernels.  Each entry point cites the reference interface it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross the boundary;
 *   - every function returns 0 on success, a negative SMK_E* code otherwise, and
 *     smk_last_error() returns a human readable message for the calling thread;
 *   - device pointers are raw HIP device addresses (torch: tensor.data_ptr());
 *   - all device work is enqueued asynchronously on the caller's hipStream_t (passed as
 *     void*; torch: torch.cuda.current_stream().cuda_stream); nothing synchronises;
 *   - the caller owns all I/O buffers; the library owns packed weights and the activation
 *     arena inside the opaque smk_ctx;
 *   - one ctx per (device, stream of use); not thread-safe, not re-entrant (the reference
 *     model is stateful in the same way: self.zf / self.feature / self.corr_feature).
 *   - tensors at the boundary are float32, NCHW, contiguous -- exactly what the reference's
 *     callers hand over / consume (tools/test.py:155,201-207,257-261).
 */
#ifndef SIAMMASK_HIP_H
#define SIAMMASK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smk_ctx smk_ctx;

/* arithmetic type of activations / weights on the device (accumulation is always fp32) */
#define SMK_DTYPE_F32 0
#define SMK_DTYPE_F16 1

/* network variant = which reference experiment's Custom is being replaced */
#define SMK_VARIANT_RPN   0   /* experiments/siamrpn_resnet/custom.py:81-93  */
#define SMK_VARIANT_BASE  1   /* experiments/siammask_base/custom.py:93-112   */
#define SMK_VARIANT_SHARP 2   /* experiments/siammask_sharp/custom.py:162-190 */

/* smk_track flags */
#define SMK_TRACK_BOX   0     /* Custom.track: cls + loc only                              */
#define SMK_TRACK_MASK  1     /* Custom.track_mask: also corr_feature (+ 63x63 mask head)  */
#define SMK_TRACK_NO_MASK_HEAD 2 /* with SMK_TRACK_MASK: skip the 3969-channel mask head
                                    (its result is never read when track_refine is used,
                                    tools/test.py:256-258); mask_out may then be NULL      */

/* error codes */
#define SMK_OK            0
#define SMK_E_ARG        -1   /* bad argument (null pointer, batch out of range, ...)      */
#define SMK_E_STATE      -2   /* call order violated (track before template, ...)          */
#define SMK_E_WEIGHT     -3   /* unknown / missing / mis-shaped weight                     */
#define SMK_E_HIP        -4   /* a HIP runtime call failed                                 */
#define SMK_E_NODEVICE   -5   /* no gfx950 device visible                                  */

/* library/ABI version: major<<16 | minor */
int smk_version(void);

/* message describing the last failure on this thread ("" if none) */
const char *smk_last_error(void);

/* ---- lifetime ------------------------------------------------------------------------
 * Replaces: Custom.__init__ + model.eval().to(device)  (tools/test.py:559-569).
 * max_batch = number of streams tracked in lock-step by this ctx (activation arena size). */
int smk_create(smk_ctx **out, int device, int dtype, int variant, int max_batch);
int smk_destroy(smk_ctx *ctx);

/* ---- weights ---------------------------------------------------------------------------
 * Replaces: utils/load_helper.py:30-54 load_pretrain -> model.load_state_dict.
 * `name` is the reference state-dict key (e.g.
 * "features.features.layer3.0.downsample.0.weight"); `data` is HOST float32, contiguous,
 * in torch layout (conv: [Cout,Cin,kh,kw]; ConvTranspose2d: [Cin,Cout,kh,kw]).
 * BatchNorm running statistics are ordinary entries; num_batches_tracked is ignored.
 * smk_finalize_weights folds BN (eval semantics, eps 1e-5), repacks to the MFMA layout,
 * uploads, and fails with SMK_E_WEIGHT naming the first missing entry. */
int smk_set_weight(smk_ctx *ctx, const char *name, const float *data,
                   const int64_t *shape, int ndim);
int smk_finalize_weights(smk_ctx *ctx);

/* ---- packed-weight cache (replaces re-running utils/load_helper.py:30-54 +
 * BN fold + repack on every start) ---------------------------------------------------------
 * smk_export_packed serialises the BN-folded, MFMA-packed weights of a finalized ctx into one
 * host blob (smk_packed_size bytes); smk_import_packed uploads such a blob instead of
 * smk_set_weight* + smk_finalize_weights.  The blob records ABI version, dtype, variant and the
 * packing constants; a mismatch is SMK_E_WEIGHT.  The CALLER keys the blob on the checkpoint
 * (siammask_amd/custom.py hashes the state dict). */
int smk_packed_size(smk_ctx *ctx, uint64_t *bytes);
int smk_export_packed(smk_ctx *ctx, void *host_buf, uint64_t capacity);
int smk_import_packed(smk_ctx *ctx, const void *host_buf, uint64_t bytes);

/* ---- the four reference methods --------------------------------------------------------
 * smk_template  <- Custom.template(template)          custom.py:173-174
 *   z_dev: [B,3,127,127] f32 NCHW, raw 0..255 BGR (tools/test.py:61-64,152-155).
 *   Caches zf and the three conv_kernel(zf) tensors (models/rpn.py:64) inside ctx.
 * smk_track     <- Custom.track / Custom.track_mask    custom.py:176-186
 *   x_dev: [B,3,255,255]; cls_out [B,10,25,25], loc_out [B,20,25,25],
 *   mask_out [B,3969,25,25] (flags & SMK_TRACK_MASK) -- f32 NCHW device buffers.
 *   B must equal the template batch (models/rpn.py:33: kernel batch defines the groups).
 * smk_refine    <- Custom.track_refine(pos)            custom.py:188-190
 *   pos_yx: B pairs (y,x), 0 <= y,x < 25; host memory if pos_on_device == 0 else device.
 *   (The reference takes ONE (y,x) for the whole batch; pass it B times for that.)
 *   out: [B,16129] f32.  Requires a preceding smk_track with SMK_TRACK_MASK. */
int smk_template(smk_ctx *ctx, const float *z_dev, int batch, void *stream);
int smk_track(smk_ctx *ctx, const float *x_dev, int batch, int flags,
              float *cls_out, float *loc_out, float *mask_out, void *stream);
int smk_refine(smk_ctx *ctx, const int32_t *pos_yx, int pos_on_device, int batch,
               float *out, void *stream);

/* ---- on-device decode (additive: the tools keep using cls/loc as before) ----
 * smk_decode restates the host code of tools/test.py:205-254 per stream on the device:
 * softmax foreground score, anchor decode (utils/anchors.py:28-51), scale/ratio penalty, cosine
 * window, argmax (lowest index wins ties, like np.argmax), with the tool's own precision stages
 * (NumPy >= 2): float32 softmax / anchor decode / exp / sz() / w-h ratio (:205-220), float64 from
 * the first division by the float64 target-size scalars on (:231-238).  Pinned against the
 * unchanged tool (tests/golden/tracker_*.npz, oracle/make_tracker_golden.py).
 *   target_wh [B,2] device f64: target size in crop pixels (w,h) = target_sz * scale_x (:230)
 *   pos_out   [B,2] device int32 (y,x) = unravel_index(best,(5,25,25))[1:] (:253-254); NULL =
 *             only the ctx-internal position used by a following smk_refine/smk_step is set
 *   box_out   [B,8] device f64: cx, cy, w, h in crop pixels (delta[:,best], :209-212; float32
 *             values), score (float32 value), penalty, pscore (float64), best_id
 * smk_set_decode_params: anchor (w,h) pairs (5), stride, hp penalty_k / window_influence
 * (config_davis.json); defaults are the reference's config.
 * smk_step = smk_track + smk_decode + smk_refine(at the decoded positions) as ONE captured
 * graph: no host round trip inside a frame.  refine_out may be NULL (no Refine). */
int smk_set_decode_params(smk_ctx *ctx, const float *anchor_wh, int n_anchor, int stride,
                          double penalty_k, double window_influence);
int smk_decode(smk_ctx *ctx, const float *cls_dev, const float *loc_dev, int batch,
               const double *target_wh_dev, int32_t *pos_out_dev, double *box_out_dev, void *stream);
int smk_step(smk_ctx *ctx, const float *x_dev, int batch, int flags, const double *target_wh_dev,
             float *cls_out, float *loc_out, float *mask_out, double *box_out, float *refine_out,
             void *stream);

/* persistent per-XCD convolution sequences (fp16, batch 8: ResNet layer2 / layer3 / adjust run as ONE conv_seq_kernel launch,
 * one workgroup per CU, image b on XCD b % 8).  The kernel needs every workgroup resident at once; when that fails (a
 * neighbour that holds CUs for more than 0.2 s, a second persistent kernel beside it, an uneven XCD placement) it raises a
 * flag in host-mapped memory, abandons the remaining layers, and the NEXT entry point called on the context (smk_template /
 * smk_track / smk_refine / smk_step / smk_seq_status) returns SMK_E_HIP: the results enqueued since then are invalid,
 * sequences are switched off for the context (per-layer kernels from there on) and the caller re-submits the frame.
 * smk_seq_status synchronises the device and reports: grid_out = workgroups per launch (0: sequences are off -- the
 * placement / occupancy check at smk_create failed, or a failure was reported); err_out = last failure (0 none,
 * 1 placement violated, 2 barrier time-out).  Returns non-zero when a failure has been reported. */
int smk_seq_status(smk_ctx *ctx, int *grid_out, int *err_out);

/* capture the launch sequences into hipGraphs and replay them (on by default when the
 * environment variable SMK_GRAPH is not "0"); graphs are keyed on (entry, batch, flags,
 * I/O pointers), so keep the I/O buffers stable to hit the cache. */
int smk_set_graph_mode(smk_ctx *ctx, int enable);

/* process-wide tuning knobs for A/B measurements (affect subsequently launched / captured work):
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
 * Environment: SMK_CHAIN_CLK=1 makes eager (non-graph) runs print the time workgroup 0 spends in each layer of
 * refine_chain_kernel to stderr (measurement aid). */
int smk_tune(const char *key, int value);
/* current value of a knob (tests and A/B scripts restore what they changed; also how a caller reads the defaults) */
int smk_tune_get(const char *key, int *value);

/* per-launch profiling: with enable != 0 every kernel launch is bracketed by HIP events on the
 * stream it is launched on (graph replay is bypassed while profiling).  smk_profile_dump
 * synchronises the device and writes a JSON array, one object per layer id in launch order:
 *   {"id","kernel","calls","ms" (sum of event durations),"flop","bytes"} where flop/bytes are
 * the ALGORITHMIC work of those launches (2*M*N*K; tensors read/written once), then resets. */
int smk_profile(smk_ctx *ctx, int enable);
int smk_profile_dump(smk_ctx *ctx, char *json_buf, int capacity);

/* read back an internal activation as f32 NCHW into a device buffer (parity tests only).
 * names: "p0","p1","p2","p3","search","zf","zk","xs","corr","head0"; batch = last batch.
 * *numel_out receives C*H*W per item; dst may be NULL to query the shape (c,h,w). */
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
 * Synchronises the stream (test helper). */
typedef struct smk_seq_op {
    smk_conv_geom g;
    int src, res_src, sync, cfg, kstag;
    const float *w_host, *b_host;
    float *y_dev;
} smk_seq_op;
int smk_op_conv_seq(const smk_seq_op *ops, int n, const float *x_dev, int iters, float *usec_out,
                    float *clk_us_out, void *stream);
int smk_op_dw_xcorr(int dtype, const float *x_dev, const float *k_dev, int B, int C,
                    int H, int W, int kh, int kw, float *y_dev, void *stream);
int smk_op_maxpool3x3s2(int dtype, const float *x_dev, int B, int C, int H, int W,
                        float *y_dev, void *stream);

/* ---- image ops either side of the network (additive) -----------------
 * smk_crop_resize <- tools/test.py:67-110 get_subwindow_tracking for B streams.
 *   frames_dev: uint8 [H][W][3] (as cv2.imread gives it); frame_stride_bytes = 0 when all streams
 *   crop the same frame (multi-object), else the byte distance between per-stream frames.
 *   boxes (host): B x (xmin, ymin, sz) = the integer window of tools/test.py:74-86 in un-padded
 *   frame coordinates (it may stick out of the frame); avg_bgr (host): B x 3 uint8 mean colour.
 *   out_dev: f32 [B,3,model_sz,model_sz].  Resize = cv2.resize INTER_LINEAR on uint8.
 * smk_paste_mask  <- tools/test.py:257-284: sigmoid(logits [B,ms*ms]) -> crop_back (cv2.warpAffine,
 *   INTER_LINEAR, BORDER_CONSTANT `border`) into a W x H frame -> (prob > seg_thr) as uint8.
 *   inv_map (host): B x 6 doubles, the INVERSE (dst -> src) affine map cv2.warpAffine derives from
 *   crop_back's mapping.  mask_out_dev [B,H,W] uint8 and/or prob_out_dev [B,H,W] f32. */
int smk_crop_resize(const uint8_t *frames_dev, int64_t frame_stride_bytes, int H, int W,
                    const int32_t *boxes, const uint8_t *avg_bgr, int B, int model_sz,
                    float *out_dev, void *stream);
int smk_paste_mask(const float *logits_dev, int mask_size, const double *inv_map, int B, int W,
                   int H, float seg_thr, float border, uint8_t *mask_out_dev, float *prob_out_dev,
                   void *stream);
/* multi-object fusion of tools/test.py:521-523 in the same pass: labels [H,W] uint8 =
 * (argmax_o prob_o + 1) * (max_o prob_o > seg_thr) over n_obj objects that share the frame */
int smk_paste_labels(const float *logits_dev, int mask_size, const double *inv_map, int n_obj, int W,
                     int H, float seg_thr, float border, uint8_t *labels_out_dev, void *stream);

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
 * smk_tune knobs -- *kernel = 0 conv_igemm_kernel, 1 conv3x3_halo_kernel, 2 conv_wreg_kernel; *bm x *bn = workgroup shape;
 * *seq_cfg = tile code inside a persistent per-XCD sequence (0 64x256, 1 64x128, 2 64x64, 3 128x256, 4 128x128) or -1 when
 * the layer cannot be part of one.  Lets the CPU test-suite pin the measured layer rules (profiles/r02_producer_waves_*). */
int smk_host_plan_conv(const smk_conv_geom *g, int dtype, int with_res, int *kernel, int *bm, int *bn, int *seq_cfg);

#ifdef __cplusplus
}
#endif
#endif /* SIAMMASK_HIP_H */
