"""GPU parity of the image ops either side of the network, called through
the C ABI (smk_crop_resize / smk_paste_mask) against oracle/cv_ops.py.
  crop + resize : uint8 arithmetic -> bit-exact (identity, exact-2x and general bilinear windows,
                  windows hanging over every frame edge, per-stream frames and a shared frame).
  paste-back    : float32; the fixed-point coordinates and tap weights are bit-identical, the only
                  difference is expf vs numpy's float32 exp inside the sigmoid: <= 5e-7 on the warped
                  probability, and the thresholded mask may differ only where |prob - thr| <= 5e-7."""
import numpy as np
import pytest
import torch

from oracle import cv_ops as C

pytestmark = pytest.mark.gpu


def _img(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(xx / 17.0) * np.cos(yy / 23.0)
    im = base[:, :, None] + rng.normal(0, 25, size=(h, w, 3))
    return np.clip(im, 0, 255).astype(np.uint8)


CASES = [  # (x, y), original_sz
    ((160.3, 120.7), 200), ((5.0, 7.5), 150), ((318.2, 239.0), 301), ((100.5, 50.5), 127), ((160.0, 120.0), 510),
    ((10.2, 230.9), 255), ((-20.0, 400.0), 181), ((200.0, 100.0), 254), ((77.7, 33.3), 91), ((250.0, 200.0), 640),
]


@pytest.mark.parametrize("model_sz", [127, 255])
def test_crop_resize_bit_exact(model_sz):
    from siammask_amd import preproc
    rng = np.random.default_rng(7)
    im = _img(rng, 240, 320)
    avg = im.mean(axis=(0, 1))
    imd = torch.from_numpy(im).cuda()
    # one launch for all windows on the shared frame
    got = preproc.crop_batch(imd, [c[0] for c in CASES], model_sz, [c[1] for c in CASES], [avg] * len(CASES)).cpu().numpy()
    for i, (pos, osz) in enumerate(CASES):
        want = C.get_subwindow_tracking(im, pos, model_sz, osz, avg)
        assert np.array_equal(got[i], want), "window %d %s sz %d -> %d: max diff %g" % (
            i, pos, osz, model_sz, np.abs(got[i] - want).max())
    # the reference's single-window signature
    one = preproc.get_subwindow_tracking(imd, CASES[0][0], model_sz, CASES[0][1], avg).cpu().numpy()
    assert np.array_equal(one, C.get_subwindow_tracking(im, CASES[0][0], model_sz, CASES[0][1], avg))


def test_crop_resize_per_stream_frames_and_many_streams():
    from siammask_amd import preproc
    rng = np.random.default_rng(8)
    B = 40                                   # > CROP_MAX_B: exercises the chunked launch
    frames = np.stack([_img(rng, 120, 160) for _ in range(B)])
    pos = [(rng.uniform(-10, 170), rng.uniform(-10, 130)) for _ in range(B)]
    osz = [int(rng.integers(60, 300)) for _ in range(B)]
    avg = [f.mean(axis=(0, 1)) for f in frames]
    got = preproc.crop_batch(torch.from_numpy(frames).cuda(), pos, 127, osz, avg).cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], C.get_subwindow_tracking(frames[b], pos[b], 127, osz[b], avg[b])), b


def test_crop_feeds_the_tracker_without_a_host_copy():
    """the crop is exactly what Custom.template / track accept (float32 NCHW, raw 0..255)"""
    from siammask_amd import preproc
    rng = np.random.default_rng(9)
    imd = torch.from_numpy(_img(rng, 240, 320)).cuda()
    z = preproc.crop_batch(imd, [(150.0, 110.0)], 127, [140], [[100.0, 110.0, 120.0]])
    assert z.shape == (1, 3, 127, 127) and z.dtype == torch.float32 and z.is_cuda
    assert float(z.min()) >= 0 and float(z.max()) <= 255 and torch.equal(z, z.round())


def test_paste_mask_matches_warp_affine_restatement():
    from siammask_amd import preproc
    rng = np.random.default_rng(10)
    B, W, H = 3, 320, 240
    yy, xx = np.mgrid[0:127, 0:127]
    logits = np.stack([6.0 * np.sin(xx / (9.0 + b)) * np.cos(yy / (7.0 + b)) + rng.normal(0, 0.5, size=(127, 127))
                       for b in range(B)]).astype(np.float32)
    crop_boxes = [[60.0, 30.0, 180.0, 180.0], [-40.5, 20.25, 333.0, 333.0], [200.0, 150.0, 90.0, 90.0]]
    deltas = [(12, 12), (3, 20), (24, 0)]
    bbs = [C.back_box(cb, d, (W, H)) for cb, d in zip(crop_boxes, deltas)]
    mask, prob = preproc.paste_masks(torch.from_numpy(logits.reshape(B, -1)).cuda(), bbs, (W, H), seg_thr=0.35,
                                     want_prob=True)
    mask, prob = mask.cpu().numpy(), prob.cpu().numpy()
    for b in range(B):
        wm, wp = C.paste_mask(logits[b], bbs[b], (W, H), 0.35)
        assert np.abs(prob[b] - wp).max() <= 5e-7, (b, np.abs(prob[b] - wp).max())
        diff = mask[b] != wm
        assert np.all(np.abs(wp[diff] - 0.35) <= 5e-7), "mask differs away from the threshold"
        assert 0 < wm.sum() < wm.size


def test_paste_labels_multi_object_fusion():
    """tools/test.py:521-523 fused into the paste-back kernel"""
    from siammask_amd import preproc
    rng = np.random.default_rng(11)
    O, W, H = 3, 200, 150
    yy, xx = np.mgrid[0:127, 0:127]
    logits = np.stack([5.0 * np.cos((xx - 40 * o) / 15.0) * np.cos((yy - 30 * o) / 13.0) + rng.normal(0, 0.3, (127, 127))
                       for o in range(O)]).astype(np.float32)
    bbs = [C.back_box([20.0 + 30 * o, 10.0 + 20 * o, 150.0, 150.0], (12, 12), (W, H)) for o in range(O)]
    lab = preproc.paste_labels(torch.from_numpy(logits.reshape(O, -1)).cuda(), bbs, (W, H), 0.35).cpu().numpy()
    probs = np.stack([C.paste_mask(logits[o], bbs[o], (W, H), 0.35)[1] for o in range(O)])
    want = ((np.argmax(probs, axis=0).astype("uint8") + 1) * (np.max(probs, axis=0) > 0.35).astype("uint8"))
    diff = lab != want
    # only pixels where two objects tie or the maximum sits on the threshold (within the exp ulp) may differ
    srt = np.sort(probs, axis=0)
    near = (np.abs(srt[-1] - 0.35) <= 5e-7) | (np.abs(srt[-1] - srt[-2]) <= 5e-7)
    assert np.all(near[diff]) and diff.mean() < 1e-3
    assert set(np.unique(want)) >= {0, 1, 2, 3}


def test_preproc_rejects_cpu_tensors():
    from siammask_amd import preproc
    with pytest.raises(RuntimeError):
        preproc.crop_batch(torch.zeros((10, 10, 3), dtype=torch.uint8), [(5, 5)], 127, [9], [[0, 0, 0]])
