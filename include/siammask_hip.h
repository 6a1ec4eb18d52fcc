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
/* ABI 1.6: split-operand fp16.  Every value of the track path's trunk (stem .. cls / loc head.3) is a pair of fp16 planes hi + lo and a
 * product the three MFMA products hi*hi + hi*lo + lo*hi in the fp32 accumulator: fp32-grade cls / loc / box -- the argmax box index of the
 * fp64 reference (/root/reference/tools/test.py:237) on every stream of tests/golden/argmax_oracle_1024.npz -- on the fp16 matrix pipe.
 * The mask head and Refine run in plain fp16 on the hi planes (their gates are the fp16 context's). */
#define SMK_DTYPE_F16X3 2

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
#define SMK_E_SEQ        -6   /* the persistent sequence kernel reported a failure (ABI 1.5;
                                 SMK_E_HIP before): frames enqueued since are invalid, the
                                 context has switched to per-layer kernels, re-submit      */

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

/* Result ring (additive; the multi-GPU flow of SURVEY.md 8e gathers boxes / masks "only at the end of a batch of frames",
 * tools/test.py:296-311 keeps them per frame): with a ring set, every smk_step ends with ONE small launch that stores the
 * frame's decoded box [batch][8] f64 and its Refine logits [batch][127*127] as fp16 into row (frames committed % rows) of the
 * caller's device buffers box_ring [rows][batch][8] / refine_ring_f16 [rows][batch][127*127] and advances a device-side frame
 * counter -- part of the captured graph, no host work, no per-frame copies by the caller.  `batch` is the batch the rows were
 * sized for (ABI 1.5: recorded; an smk_step with another batch fails with SMK_E_ARG instead of writing past the rows).
 * refine_ring_f16 may be NULL (boxes only); rows = 0 switches the ring off.  Synchronises the device
 * and drops the captured graphs.  smk_result_ring_cursor synchronises `stream` (behind an outstanding pipelined tail), returns
 * the number of frames committed (mod 2^32) and optionally resets it.
 * A frame whose persistent sequence launch FAILED (SMK_E_SEQ from the next entry point / smk_seq_sync_check) has still committed a
 * row -- of invalid values -- and advanced the counter: the caller that re-submits the frame either resets the counter
 * (smk_result_ring_cursor(.., reset = 1)) and re-runs from the last frame it trusts, or overwrites by position: the re-submitted frame
 * lands in the NEXT row.  The library does not rewind the cursor (rows may already have been handed to a gather). */
int smk_set_result_ring(smk_ctx *ctx, double *box_ring_dev, void *refine_ring_f16_dev, int rows, int batch);
int smk_result_ring_cursor(smk_ctx *ctx, int *frames_out, int reset, void *stream);

/* Software-pipelined frame steps (ABI 1.5, additive).  The reference's tracker needs only the decoded box of frame f to crop
 * frame f + 1 (tools/test.py:240-250,302-308); the Refine mask (:257-284) is an output.  smk_set_pipeline(ctx, 1) lets smk_step
 * use that: a step with refine_out enqueues
 *     on `stream`:       stem + layer1 of frame f | wait for the tail of frame f - 1 | layer2 .. heads .. decode of frame f
 *     on a side stream:  the Refine module (+ the 63x63 mask head) of frame f at the decoded positions
 * so that the small, low-occupancy launches of the tail share the chip with the bandwidth-bound front end of the next frame.
 * Contract: cls / loc / box_out (and the ring's box row) of frame f are complete in `stream` order as before; mask_out /
 * refine_out (and the ring's logits row + cursor) of frame f are complete once the NEXT smk_step's decode is, or behind
 * smk_pipeline_join(ctx, any_stream), which orders that stream behind the outstanding tail (every other entry point of the
 * context joins implicitly).  Results are bit-identical to the serial step.  Costs a second copy of p0 / p1 (4 MB per stream of the
 * batch in fp16).  depth 0 = serial (default).  Synchronises the device.
 * depth 2 (throughput mode; fp16 contexts, batches that run the persistent sequence; otherwise it behaves like depth 1): the tail is
 * cut in two -- the window convolutions + deconv + v*.2 run beside the next frame's front end as in depth 1, the Refine chain + mask
 * head (one low-occupancy launch) wait until the NEXT frame's persistent launch has left and run beside that frame's heads.  That
 * second part is launched by the next smk_step; smk_pipeline_join (and every other entry point) launches it at once.  mask_out /
 * refine_out of frame f are then complete behind smk_pipeline_join only (or one smk_step later: smk_pipeline_observe orders a stream
 * behind what has been launched so far WITHOUT launching a pending second part).  Same bits.  One more copy of head0. */
int smk_set_pipeline(smk_ctx *ctx, int depth);
int smk_pipeline_join(smk_ctx *ctx, void *stream);
int smk_pipeline_observe(smk_ctx *ctx, void *stream);

/* persistent per-XCD convolution sequences (fp16, batch 8: ResNet layer2 / layer3 / adjust run as ONE conv_seq_kernel launch,
 * one workgroup per CU, image b on XCD b % 8).  The kernel needs every workgroup resident at once; when that fails (a
 * neighbour that holds CUs for more than 0.2 s, a second persistent kernel beside it, an uneven XCD placement) it raises a
 * flag in host-mapped memory, abandons the remaining layers, and smk_seq_sync_check behind the call -- or, for callers that do
 * not use it, the NEXT entry point called on the context (smk_template / smk_track / smk_refine / smk_step /
 * smk_seq_status) -- returns SMK_E_SEQ: the results enqueued since then are invalid,
 * sequences are switched off for the context (per-layer kernels from there on) and the caller re-submits the frame.
 * smk_seq_status synchronises the device and reports: grid_out = workgroups per launch (0: sequences are off -- the
 * placement / occupancy check at smk_create failed, or a failure was reported); err_out = last failure (0 none,
 * 1 placement violated, 2 barrier time-out).  Returns non-zero when a failure has been reported. */
int smk_seq_status(smk_ctx *ctx, int *grid_out, int *err_out);
/* The same check scoped to ONE call: when a sequence launch has been enqueued on the context since the flag was last looked at,
 * synchronise `stream` (the stream the entry points were given) and read the flag -- a failure is returned by the call that
 * produced the invalid frame, not by the next one.  Costs nothing (no synchronisation) when no sequence launch is pending.
 * Callers that read results right behind the call (the reference's tools do: tools/test.py:205 `.cpu()`) call it where they
 * would synchronise anyway; siammask_amd.custom does so in template / track / track_mask / track_refine and re-runs the frame
 * on the per-layer kernels.  synced_out (may be NULL): 1 when the stream was synchronised. */
int smk_seq_sync_check(smk_ctx *ctx, void *stream, int *synced_out);

/* capture the launch sequences into hipGraphs and replay them (on by default when the
 * environment variable SMK_GRAPH is not "0"); graphs are keyed on (entry, batch, flags,
 * I/O pointers), so keep the I/O buffers stable to hit the cache. */
int smk_set_graph_mode(smk_ctx *ctx, int enable);

/* Tuning knobs, per-launch profiling, the read-back of internal activations, the per-op entry points the parity tests
 * use and the measurement aids live in siammask_hip_test.h (same library, not needed to drive the model). */

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

#ifdef __cplusplus
}
#endif
#endif /* SIAMMASK_HIP_H */
