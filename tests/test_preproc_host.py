"""Host-side geometry of the image ops either side of the network (no GPU needed): the integer crop window,
the crop_back affine map and its inverse, and the mask paste-back box, each against the reference's literal
arithmetic (tools/test.py, cited per test)."""
import numpy as np

from siammask_amd import preproc
from siammask_amd.tracker import TrackerConfig, preproc_back_box


def test_subwindow_box_is_the_reference_window():
    """tools/test.py:70-76: c = (original_sz+1)/2; context_xmin = round(pos[0]-c); xmax = xmin + sz - 1"""
    rng = np.random.default_rng(0)
    for _ in range(500):
        pos = rng.uniform(-50, 900, size=2)
        sz = int(rng.integers(20, 700))
        c = (sz + 1) / 2
        xmin, ymin = round(pos[0] - c), round(pos[1] - c)          # Python round == the tool's round()
        bx = preproc.subwindow_box(pos, sz)
        assert bx == (int(xmin), int(ymin), sz)
        assert bx[0] + sz - 1 == int(xmin + sz - 1)                 # context_xmax


def test_invert_affine_inverts():
    """cv2.warpAffine is handed the forward map and inverts it (cv::invertAffineTransform)"""
    rng = np.random.default_rng(1)
    for _ in range(200):
        m = np.array([[rng.uniform(0.2, 5), rng.uniform(-0.5, 0.5), rng.uniform(-300, 300)],
                      [rng.uniform(-0.5, 0.5), rng.uniform(0.2, 5), rng.uniform(-300, 300)]])
        inv = preproc.invert_affine(m).reshape(2, 3)
        f = np.vstack([m, [0, 0, 1]])
        g = np.vstack([inv, [0, 0, 1]])
        assert np.allclose(f @ g, np.eye(3), atol=1e-9)
    # singular input: OpenCV returns the zero matrix
    assert np.all(preproc.invert_affine(np.zeros((2, 3))) == 0)


def test_crop_back_map_is_the_reference_mapping():
    """tools/test.py:263-268: a = (out_w-1)/bbox[2]; b = (out_h-1)/bbox[3]; c = -a*bbox[0]; d = -b*bbox[1]"""
    rng = np.random.default_rng(2)
    for _ in range(200):
        bbox = [rng.uniform(-200, 50), rng.uniform(-200, 50), rng.uniform(50, 900), rng.uniform(50, 900)]
        out_sz = (int(rng.integers(100, 1300)), int(rng.integers(100, 800)))
        a = (out_sz[0] - 1) / bbox[2]
        b = (out_sz[1] - 1) / bbox[3]
        c = -a * bbox[0]
        d = -b * bbox[1]
        ref = np.array([[a, 0, c], [0, b, d]]).astype(np.float64)
        assert np.array_equal(preproc.crop_back_map(bbox, out_sz), ref)


def test_paste_back_box_is_the_reference_box():
    """tools/test.py:275-279 with p.out_size = 127 (refine) and 63 (base head)"""
    p = TrackerConfig()
    rng = np.random.default_rng(3)
    for mask_size in (127, 63):
        for _ in range(200):
            crop_box = [rng.uniform(-100, 600), rng.uniform(-100, 400), rng.uniform(60, 500)]
            dy, dx = int(rng.integers(0, 25)), int(rng.integers(0, 25))
            im_w, im_h = int(rng.integers(200, 1300)), int(rng.integers(200, 800))
            s = crop_box[2] / p.instance_size
            sub_box = [crop_box[0] + (dx - p.base_size / 2) * p.total_stride * s,
                       crop_box[1] + (dy - p.base_size / 2) * p.total_stride * s,
                       s * p.exemplar_size, s * p.exemplar_size]
            s2 = mask_size / sub_box[2]
            ref = [-sub_box[0] * s2, -sub_box[1] * s2, im_w * s2, im_h * s2]
            got = preproc_back_box(crop_box, (dy, dx), (im_w, im_h), p, mask_size)
            assert np.allclose(got, ref, rtol=0, atol=0)


def test_device_ops_refuse_cpu_tensors():
    """no CPU fallback: the image ops raise on CPU tensors instead of computing something else"""
    import pytest
    import torch
    with pytest.raises(RuntimeError):
        preproc.crop_batch(torch.zeros((8, 8, 3), dtype=torch.uint8), [(4, 4)], 255, [8], [(0, 0, 0)])
    with pytest.raises(RuntimeError):
        preproc.paste_masks(torch.zeros((1, 127 * 127)), [[0, 0, 10, 10]], (10, 10))
