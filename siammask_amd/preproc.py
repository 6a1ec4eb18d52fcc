"""Device-side image ops either side of the network, on top of
libsiammask_hip.so.  Additive: tools/test.py keeps its numpy/cv2 versions; these take the same
arguments but a frame that already lives on the MI355X.

  get_subwindow_tracking <- tools/test.py:67-110   (crop + mean-colour pad + cv2.resize INTER_LINEAR)
  crop_batch              the same for B streams in one launch
  paste_masks            <- tools/test.py:257-284  (sigmoid + crop_back/cv2.warpAffine + threshold)
No CPU fallback: CPU tensors raise."""
import ctypes

import numpy as np
import torch

from . import _lib


def _need_cuda(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("siammask_amd.preproc runs on the MI355X only: %s must be a CUDA(HIP) tensor" % what)


def subwindow_box(pos, original_sz):
    """tools/test.py:70-76: integer crop window (xmin, ymin, sz) in un-padded frame coordinates."""
    c = (original_sz + 1) / 2
    return int(round(float(pos[0]) - c)), int(round(float(pos[1]) - c)), int(original_sz)


def crop_batch(frames, positions, model_sz, original_szs, avg_chans):
    """frames: uint8 CUDA tensor [H,W,3] (one frame shared by all streams) or [B,H,W,3];
    positions: B x (x, y) window centres; original_szs: B window sizes (tools/test.py passes
    round(s_x)); avg_chans: B x 3 mean colours (np.mean(im, axis=(0,1))).
    -> float32 CUDA tensor [B,3,model_sz,model_sz] (what im_to_torch + stacking would give)."""
    _need_cuda(frames, "frames")
    if frames.dtype != torch.uint8 or frames.dim() not in (3, 4) or frames.shape[-1] != 3:
        raise ValueError("frames must be uint8 [H,W,3] or [B,H,W,3]")
    frames = frames.contiguous()
    B = len(positions)
    if frames.dim() == 4 and frames.shape[0] != B:
        raise ValueError("frames batch %d != %d positions" % (frames.shape[0], B))
    H, W = int(frames.shape[-3]), int(frames.shape[-2])
    stride = H * W * 3 if frames.dim() == 4 else 0
    boxes = np.asarray([subwindow_box(p, s) for p, s in zip(positions, original_szs)], dtype=np.int32).reshape(B, 3)
    # numpy assignment of the float mean into a uint8 image truncates (tools/test.py:92-99)
    avg = np.asarray(avg_chans, dtype=np.float64).reshape(B, 3).astype(np.uint8)
    out = torch.empty((B, 3, model_sz, model_sz), dtype=torch.float32, device=frames.device)
    with torch.cuda.device(frames.device):
        _lib.check(_lib.lib().smk_crop_resize(
            frames.data_ptr(), stride, H, W, boxes.ctypes.data_as(ctypes.c_void_p),
            np.ascontiguousarray(avg).ctypes.data_as(ctypes.c_void_p), B, int(model_sz), out.data_ptr(),
            _lib.current_stream_ptr()))
    return out


def get_subwindow_tracking(im, pos, model_sz, original_sz, avg_chans, out_mode="torch"):
    """Same arguments as tools/test.py:67 with ``im`` a uint8 CUDA tensor [H,W,3].
    -> float32 CUDA tensor [3,model_sz,model_sz] (out_mode 'torch')."""
    if out_mode not in "torch":
        raise NotImplementedError("only out_mode='torch' (the tracker's use, tools/test.py:154,198)")
    return crop_batch(im, [pos], model_sz, [original_sz], [avg_chans])[0]


def invert_affine(mapping):
    """cv::invertAffineTransform in float64 (cv2.warpAffine inverts the forward map it is given)."""
    m = np.asarray(mapping, dtype=np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([a11, a12, b1, a21, a22, b2], dtype=np.float64)


def crop_back_map(bbox, out_sz):
    """the forward mapping of crop_back (tools/test.py:263-268)"""
    a = (out_sz[0] - 1) / bbox[2]
    b = (out_sz[1] - 1) / bbox[3]
    return np.array([[a, 0, -a * bbox[0]], [0, b, -b * bbox[1]]], dtype=np.float64)


def paste_masks(logits, back_boxes, im_wh, seg_thr=0.35, padding=-1.0, want_prob=False):
    """logits: float32 CUDA tensor [B, ms*ms] (track_refine output); back_boxes: B boxes
    (tools/test.py:279: [-sub_box[0]*s, -sub_box[1]*s, im_w*s, im_h*s]); im_wh = (im_w, im_h).
    -> uint8 CUDA tensor [B,im_h,im_w] = (crop_back(sigmoid(mask)) > seg_thr)  [, float32 prob map]."""
    _need_cuda(logits, "logits")
    logits = logits.contiguous().float()
    B = logits.shape[0]
    ms = int(round(logits[0].numel() ** 0.5))
    if ms * ms != logits[0].numel() or len(back_boxes) != B:
        raise ValueError("logits must be [B, ms*ms] with one back_box per stream")
    W, H = int(im_wh[0]), int(im_wh[1])
    inv = np.ascontiguousarray(np.stack([invert_affine(crop_back_map(bb, (W, H))) for bb in back_boxes]))
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=logits.device)
    prob = torch.empty((B, H, W), dtype=torch.float32, device=logits.device) if want_prob else None
    with torch.cuda.device(logits.device):
        _lib.check(_lib.lib().smk_paste_mask(
            logits.data_ptr(), ms, inv.ctypes.data_as(ctypes.c_void_p), B, W, H, float(seg_thr), float(padding),
            mask.data_ptr(), prob.data_ptr() if prob is not None else None, _lib.current_stream_ptr()))
    return (mask, prob) if want_prob else mask


def paste_labels(logits, back_boxes, im_wh, seg_thr=0.35, padding=-1.0):
    """Multi-object VOS fusion (tools/test.py:521-523) fused with the paste-back: the O objects of one
    frame -> uint8 label map [im_h, im_w] = (argmax_o prob_o + 1) * (max_o prob_o > seg_thr)."""
    _need_cuda(logits, "logits")
    logits = logits.contiguous().float()
    O = logits.shape[0]
    ms = int(round(logits[0].numel() ** 0.5))
    if ms * ms != logits[0].numel() or len(back_boxes) != O:
        raise ValueError("logits must be [O, ms*ms] with one back_box per object")
    W, H = int(im_wh[0]), int(im_wh[1])
    inv = np.ascontiguousarray(np.stack([invert_affine(crop_back_map(bb, (W, H))) for bb in back_boxes]))
    labels = torch.empty((H, W), dtype=torch.uint8, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(_lib.lib().smk_paste_labels(logits.data_ptr(), ms, inv.ctypes.data_as(ctypes.c_void_p), O, W, H,
                                               float(seg_thr), float(padding), labels.data_ptr(),
                                               _lib.current_stream_ptr()))
    return labels
