"""Device-resident tracker loop (siammask_amd/tracker.py: siamese_init / siamese_track of
tools/test.py:132-311 on the device ops) against the same loop restated on the CPU oracles
(oracle/np_oracle.Oracle in float64 + oracle/cv_ops + decode_best).  fp32 device path:
best anchor index identical, box / state within 1e-3, pasted mask IoU >= 0.99."""
import numpy as np
import pytest
import torch

from oracle import cv_ops as C
from oracle.np_oracle import Oracle, decode_best
from siammask_amd import synth

pytestmark = pytest.mark.gpu
HP = {"penalty_k": 0.04, "window_influence": 0.4, "lr": 1.0, "seg_thr": 0.35, "out_size": 127}


def _frame(rng, h, w, cx, cy):
    yy, xx = np.mgrid[0:h, 0:w]
    base = 110 + 60 * np.sin(xx / 19.0) * np.cos(yy / 27.0)
    blob = 90 * np.exp(-(((xx - cx) / 28.0) ** 2 + ((yy - cy) / 20.0) ** 2))
    im = base[:, :, None] + blob[:, :, None] * np.array([1.0, 0.6, 0.3]) + rng.normal(0, 6, size=(h, w, 3))
    return np.clip(im, 0, 255).astype(np.uint8)


def _oracle_step(o, im, pos, sz, avg, p):
    """siamese_track for one stream on the CPU oracles (tools/test.py:173-311)"""
    wc_x = sz[1] + p["context_amount"] * sz.sum()
    hc_x = sz[0] + p["context_amount"] * sz.sum()
    s = np.sqrt(wc_x * hc_x)
    scale_x = 127 / s
    s_x = s + 2 * ((255 - 127) / 2 / scale_x)
    r = round(s_x)
    crop_box = [pos[0] - r / 2, pos[1] - r / 2, r, r]
    x = C.get_subwindow_tracking(im, pos, 255, r, avg)[None].astype(np.float64)
    cls, loc, _ = o.track_mask(x)
    best, dy, dx, _ = decode_best(cls[0], loc[0], target_sz=(sz[0], sz[1]), scale_x=scale_x, penalty_k=p["penalty_k"],
                                  window_influence=p["window_influence"])
    box = decode_best.last["box"]
    pred = box[:4] / scale_x
    lr = box[5] * box[4] * p["lr"]
    new_pos = np.array([pred[0] + pos[0], pred[1] + pos[1]])
    new_sz = np.array([sz[0] * (1 - lr) + pred[2] * lr, sz[1] * (1 - lr) + pred[3] * lr])
    logits = o.track_refine((dy, dx))[0].reshape(127, 127)
    bb = C.back_box(crop_box, (dy, dx), (im.shape[1], im.shape[0]))
    mask, _ = C.paste_mask(logits, bb, (im.shape[1], im.shape[0]), p["seg_thr"])
    new_pos = np.array([np.clip(new_pos[0], 0, im.shape[1]), np.clip(new_pos[1], 0, im.shape[0])])
    new_sz = np.array([np.clip(new_sz[0], 10, im.shape[1]), np.clip(new_sz[1], 10, im.shape[0])])
    return best, new_pos, new_sz, box[4], mask


def test_device_tracker_loop_matches_cpu_restatement():
    from siammask_amd.custom import build
    from siammask_amd.tracker import DeviceTracker
    rng = np.random.default_rng(21)
    f0, f1 = _frame(rng, 240, 320, 150, 120), _frame(rng, 240, 320, 158, 116)
    pos0 = np.array([[150.0, 120.0], [60.0, 200.0]])                    # second stream hangs over the frame edge
    sz0 = np.array([[70.0, 50.0], [90.0, 60.0]])
    m = build("sharp", dtype="f32", max_batch=2)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    m = m.eval().cuda()
    tr = DeviceTracker(m, HP)
    tr.init(torch.from_numpy(f0).cuda(), pos0, sz0)
    st = tr.track(torch.from_numpy(f1).cuda())
    masks = st["mask"].cpu().numpy()
    p = dict(HP, context_amount=0.5)
    sd = synth.state_dict("sharp", "synthetic_damped")
    avg = f0.mean(axis=(0, 1))
    for b in range(2):
        o = Oracle(sd, "sharp")
        wc_z = sz0[b, 0] + 0.5 * sz0[b].sum()
        hc_z = sz0[b, 1] + 0.5 * sz0[b].sum()
        z = C.get_subwindow_tracking(f0, pos0[b], 127, round(np.sqrt(wc_z * hc_z)), avg)[None].astype(np.float64)
        o.template(z)
        best, npos, nsz, score, mask = _oracle_step(o, f1, pos0[b], sz0[b], avg, p)
        dy, dx = st["delta_yx"][b]
        assert (int(dy), int(dx)) == ((best % 625) // 25, best % 25), "stream %d: best anchor position differs" % b
        assert np.allclose(st["target_pos"][b], npos, rtol=0, atol=2e-3), (b, st["target_pos"][b], npos)
        assert np.allclose(st["target_sz"][b], nsz, rtol=1e-4, atol=2e-3), (b, st["target_sz"][b], nsz)
        assert abs(st["score"][b] - score) <= 1e-4
        inter, union = (masks[b] & mask).sum(), (masks[b] | mask).sum()
        assert union == 0 or inter / union >= 0.99, (b, inter, union)
