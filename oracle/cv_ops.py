"""CPU oracle for the host-side image ops around the hot path.

    *** TEST INFRASTRUCTURE ONLY *** (same rule as oracle/np_oracle.py)

Restates, in numpy:
  * get_subwindow_tracking           tools/test.py:67-110  (crop + mean-colour pad + resize)
  * crop_back / mask paste-back      tools/test.py:257-284
and the two OpenCV primitives those call, which are NOT under /root/reference:
  * cv2.resize(uint8, INTER_LINEAR)  OpenCV 3.4 (requirements.txt:8 pins opencv-python==3.4.3.18),
    modules/imgproc/src/resize.cpp: 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS),
    HResizeLinear / VResizeLinear<uchar,int,short,...>, and the exact-2x shortcut that turns
    INTER_LINEAR into the 2x2 box average (resize(): "is_area_fast && iscale == 2").
  * cv2.warpAffine(float32, INTER_LINEAR, BORDER_CONSTANT)  modules/imgproc/src/imgwarp.cpp:
    inverse map in 10-bit fixed point (AB_BITS), 1/32-pixel interpolation table (INTER_BITS = 5),
    float bilinear taps, constant border.

PARITY UNPINNED for the OpenCV primitives: cv2 is not installed in this image and cannot be
installed (no network), and the reference holds no golden vectors for them.  What IS checked
(tests/test_cv_ops.py): the sampling geometry against torch.nn.functional.interpolate /
grid_sample (same half-pixel convention) to within the fixed-point quantisation, exactness on
constant / identity / exact-2x inputs, and the padding logic against a literal re-run of the
reference's numpy code.  The rounding details follow the published OpenCV source as cited.
"""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _cv_round_f32(v):
    """cvRound(float) = lrintf: round half to even"""
    return np.rint(np.asarray(v, dtype=np.float32)).astype(np.int64)


def _linear_coeffs(ssize, dsize):
    """resize.cpp (linear): per destination index the source index and the two 11-bit weights"""
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= ssize - 1
    f[hi] = 0.0
    s[hi] = ssize - 1
    a0 = np.clip(_cv_round_f32((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE)), -32768, 32767)
    a1 = np.clip(_cv_round_f32(f * np.float32(INTER_RESIZE_COEF_SCALE)), -32768, 32767)
    s1 = np.minimum(s + 1, ssize - 1)
    return s, s1, a0, a1


def cv_resize_linear_u8(src, dsize):
    """cv2.resize(src_uint8[H,W,C], (dsize_w, dsize_h)) with the default INTER_LINEAR."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[0], src.shape[1]
    if (dw, dh) == (sw, sh):
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:
        # resize(): INTER_LINEAR with an exact 2x2 decimation is computed as INTER_AREA
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _linear_coeffs(sw, dw)
    y0, y1, ay0, ay1 = _linear_coeffs(sh, dh)
    s = src.astype(np.int64)
    # HResizeLinear: rows of ints  S[sx]*a0 + S[sx+1]*a1
    h = s[:, x0, :] * ax0[None, :, None] + s[:, x1, :] * ax1[None, :, None]
    r0, r1 = h[y0], h[y1]
    # VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>: the 8-bit specialisation
    b0, b1 = ay0[:, None, None], ay1[:, None, None]
    out = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def py_round(x):
    """Python 3 round() of a float to an int (round half to even), tools/test.py:74,76"""
    return int(round(float(x)))


def subwindow_box(pos, original_sz, im_shape):
    """tools/test.py:70-86: integer crop window [xmin, ymin, sz] in un-padded image coordinates
    (may start before 0 / end past the image: those pixels are the mean colour)."""
    sz = original_sz
    c = (original_sz + 1) / 2
    xmin = py_round(pos[0] - c)
    ymin = py_round(pos[1] - c)
    return xmin, ymin, int(sz)


def get_subwindow_tracking(im, pos, model_sz, original_sz, avg_chans):
    """tools/test.py:67-110 with out_mode='torch' semantics: -> float32 [3, model_sz, model_sz]."""
    im = np.asarray(im)
    assert im.dtype == np.uint8 and im.ndim == 3
    xmin, ymin, sz = subwindow_box(pos, original_sz, im.shape)
    H, W = im.shape[0], im.shape[1]
    ys, xs = np.arange(ymin, ymin + sz), np.arange(xmin, xmin + sz)
    inside = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
    patch = np.empty((sz, sz, im.shape[2]), dtype=np.uint8)
    patch[:] = np.asarray(avg_chans).astype(np.uint8)            # numpy assignment into uint8: truncation (:92-99)
    yy, xx = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
    vals = im[yy][:, xx]
    patch[inside] = vals[inside]
    if model_sz != sz:
        patch = cv_resize_linear_u8(patch, (model_sz, model_sz))
    return np.transpose(patch, (2, 0, 1)).astype(np.float32)     # im_to_torch (:61-64)


def get_subwindow_tracking_literal(im, pos, model_sz, original_sz, avg_chans):
    """the reference's own pad-then-slice construction (tools/test.py:70-103), kept literal, to
    check the index form above; resize through cv_resize_linear_u8"""
    sz = original_sz
    im_sz = im.shape
    c = (original_sz + 1) / 2
    context_xmin = round(pos[0] - c)
    context_xmax = context_xmin + sz - 1
    context_ymin = round(pos[1] - c)
    context_ymax = context_ymin + sz - 1
    left_pad = int(max(0., -context_xmin))
    top_pad = int(max(0., -context_ymin))
    right_pad = int(max(0., context_xmax - im_sz[1] + 1))
    bottom_pad = int(max(0., context_ymax - im_sz[0] + 1))
    context_xmin = context_xmin + left_pad
    context_xmax = context_xmax + left_pad
    context_ymin = context_ymin + top_pad
    context_ymax = context_ymax + top_pad
    r, c, k = im.shape
    if any([top_pad, bottom_pad, left_pad, right_pad]):
        te_im = np.zeros((r + top_pad + bottom_pad, c + left_pad + right_pad, k), np.uint8)
        te_im[top_pad:top_pad + r, left_pad:left_pad + c, :] = im
        if top_pad:
            te_im[0:top_pad, left_pad:left_pad + c, :] = avg_chans
        if bottom_pad:
            te_im[r + top_pad:, left_pad:left_pad + c, :] = avg_chans
        if left_pad:
            te_im[:, 0:left_pad, :] = avg_chans
        if right_pad:
            te_im[:, c + left_pad:, :] = avg_chans
        patch = te_im[int(context_ymin):int(context_ymax + 1), int(context_xmin):int(context_xmax + 1), :]
    else:
        patch = im[int(context_ymin):int(context_ymax + 1), int(context_xmin):int(context_xmax + 1), :]
    if model_sz != original_sz:
        patch = cv_resize_linear_u8(np.ascontiguousarray(patch), (model_sz, model_sz))
    return np.transpose(patch, (2, 0, 1)).astype(np.float32)


# ------------------------------------------------------------------------------------------
# cv2.warpAffine(float32, INTER_LINEAR, BORDER_CONSTANT)  +  crop_back (tools/test.py:263-274)
# ------------------------------------------------------------------------------------------
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def _sat_int(v):
    """saturate_cast<int>(double) = cvRound: round half to even"""
    return np.rint(np.asarray(v, dtype=np.float64)).astype(np.int64)


def invert_affine(m):
    """cv::invertAffineTransform (imgwarp.cpp), double precision"""
    m = np.asarray(m, dtype=np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]], dtype=np.float64)


def cv_warp_affine_linear_f32(src, mapping, dsize, border_value):
    """cv2.warpAffine(src_f32[H,W], M(2x3, forward map), (w, h), flags=INTER_LINEAR,
    borderMode=BORDER_CONSTANT, borderValue).  WarpAffineInvoker + remapBilinear<float>."""
    src = np.asarray(src, dtype=np.float32)
    sh, sw = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    M = invert_affine(mapping)                      # warpAffine inverts unless WARP_INVERSE_MAP
    x = np.arange(dw, dtype=np.float64)
    adelta = _sat_int(M[0, 0] * x * AB_SCALE)
    bdelta = _sat_int(M[1, 0] * x * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    y = np.arange(dh, dtype=np.float64)
    X0 = _sat_int((M[0, 1] * y + M[0, 2]) * AB_SCALE) + round_delta
    Y0 = _sat_int((M[1, 1] * y + M[1, 2]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)        # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx = (X & (INTER_TAB_SIZE - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB_SIZE)
    fy = (Y & (INTER_TAB_SIZE - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB_SIZE)
    # initInterTab2D: w = (1-fy|fy) x (1-fx|fx) in float
    w00 = (np.float32(1) - fy) * (np.float32(1) - fx)
    w01 = (np.float32(1) - fy) * fx
    w10 = fy * (np.float32(1) - fx)
    w11 = fy * fx
    bv = np.float32(border_value)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(ok, v, bv).astype(np.float32)

    out = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11
    return out.astype(np.float32)


def crop_back(image, bbox, out_sz, padding=-1):
    """tools/test.py:263-274"""
    a = (out_sz[0] - 1) / bbox[2]
    b = (out_sz[1] - 1) / bbox[3]
    c = -a * bbox[0]
    d = -b * bbox[1]
    mapping = np.array([[a, 0, c], [0, b, d]]).astype(np.float64)
    return cv_warp_affine_linear_f32(image, mapping, (out_sz[0], out_sz[1]), padding)


def back_box(crop_box, delta_yx, im_wh, instance_size=255, exemplar_size=127, base_size=8, total_stride=8,
             out_size=127):
    """tools/test.py:275-281: the box that maps the 127x127 refine mask into the image"""
    delta_y, delta_x = delta_yx
    s = crop_box[2] / instance_size
    sub_box = [crop_box[0] + (delta_x - base_size / 2) * total_stride * s,
               crop_box[1] + (delta_y - base_size / 2) * total_stride * s,
               s * exemplar_size, s * exemplar_size]
    s = out_size / sub_box[2]
    return [-sub_box[0] * s, -sub_box[1] * s, im_wh[0] * s, im_wh[1] * s]


def paste_mask(logits, bbox, im_wh, seg_thr=0.35):
    """tools/test.py:257-261,282-284: sigmoid -> crop_back(padding=-1) -> threshold -> uint8"""
    lg = np.asarray(logits, dtype=np.float32)
    prob = (np.float32(1) / (np.float32(1) + np.exp(-lg, dtype=np.float32))).astype(np.float32)
    m = crop_back(prob, bbox, im_wh)
    return (m > seg_thr).astype(np.uint8), m
