"""CPU checks of oracle/cv_ops.py (restatement of get_subwindow_tracking / crop_back and of the two
OpenCV primitives they call).  cv2 itself is not installable here, so the OpenCV rounding details
are unpinned (see the oracle's header); what is pinned: sampling geometry against torch, exact
cases, and the padding logic against the reference's literal numpy code."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import cv_ops as C


def test_resize_geometry_matches_torch_bilinear():
    rng = np.random.default_rng(0)
    for sh, dh in ((200, 127), (90, 127), (300, 255), (511, 255), (37, 255)):
        src = rng.integers(0, 256, size=(sh, sh + 3, 3), dtype=np.uint8)
        out = C.cv_resize_linear_u8(src, (dh, dh))
        t = torch.from_numpy(src.astype(np.float32)).permute(2, 0, 1)[None]
        ref = F.interpolate(t, size=(dh, dh), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        # 11-bit taps + the truncating 8-bit vertical pass stay within one grey level of float bilinear
        assert np.abs(out.astype(np.float64) - ref).max() < 1.0


def test_resize_exact_cases():
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, size=(254, 254, 3), dtype=np.uint8)
    assert np.array_equal(C.cv_resize_linear_u8(src, (254, 254)), src)                  # identity
    box = C.cv_resize_linear_u8(src, (127, 127))                                         # exact 2x: box average
    s = src.astype(np.int64)
    assert np.array_equal(box, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2))
    const = np.full((61, 61, 3), 201, np.uint8)
    assert np.unique(C.cv_resize_linear_u8(const, (255, 255))).tolist() == [201]         # weights sum to 2048


def test_subwindow_index_form_equals_reference_pad_and_slice():
    rng = np.random.default_rng(2)
    im = rng.integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    avg = im.mean(axis=(0, 1))
    for pos, osz in (((160.3, 120.7), 200), ((5.0, 7.5), 150), ((318.2, 239.0), 301), ((100.5, 50.5), 127),
                     ((160, 120), 510), ((10.2, 230.9), 255), ((-20.0, 400.0), 181)):
        for msz in (127, 255):
            a = C.get_subwindow_tracking(im, pos, msz, osz, avg)
            b = C.get_subwindow_tracking_literal(im, np.array(pos), msz, osz, avg)
            assert a.shape == (3, msz, msz) and np.array_equal(a, b)


def test_warp_affine_geometry_and_border():
    rng = np.random.default_rng(3)
    mask = rng.normal(size=(127, 127)).astype(np.float32)
    bbox = [-37.3, -21.9, 320 * 0.8, 240 * 0.8]
    w = C.crop_back(mask, bbox, (320, 240), -1)
    a, b = (320 - 1) / bbox[2], (240 - 1) / bbox[3]
    xs, ys = (np.arange(320) + a * bbox[0]) / a, (np.arange(240) + b * bbox[1]) / b
    X, Y = np.meshgrid(xs, ys)
    x0, y0 = np.floor(X).astype(int), np.floor(Y).astype(int)
    fx, fy = X - x0, Y - y0

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < 127) & (xx >= 0) & (xx < 127)
        return np.where(ok, mask[np.clip(yy, 0, 126), np.clip(xx, 0, 126)], -1.0)

    ref = tap(y0, x0) * (1 - fy) * (1 - fx) + tap(y0, x0 + 1) * (1 - fy) * fx + tap(y0 + 1, x0) * fy * (1 - fx) \
        + tap(y0 + 1, x0 + 1) * fy * fx
    # coordinates are quantised to 1/32 pixel: error <= gradient / 64 per axis
    g = max(np.abs(np.diff(mask, axis=0)).max(), np.abs(np.diff(mask, axis=1)).max())
    assert np.abs(w - ref).max() <= 2.2 * g / 64 + 1e-5
    # far outside the mask everything is the border value
    far = C.crop_back(mask, [500.0, 500.0, 100.0, 100.0], (64, 48), -1)
    assert np.all(far == -1)


def test_paste_mask_threshold():
    lg = np.full((127, 127), -4.0, np.float32)
    lg[40:90, 30:100] = 4.0
    m, prob = C.paste_mask(lg, C.back_box([100.0, 60.0, 220.0, 220.0], (12, 12), (640, 360)), (640, 360))
    assert m.dtype == np.uint8 and m.shape == (360, 640) and 0 < m.sum() < m.size
    assert prob.min() >= -1.0 and prob.max() <= 1.0
