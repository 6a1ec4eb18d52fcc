#!/usr/bin/env python
"""Generate tests/golden/tracker_*.npz by running the reference's UNCHANGED tools/test.py
(`siamese_init` :132-170, `siamese_track` :173-315) with the reference's own `Custom` on this container's CPU.

    python oracle/make_tracker_golden.py

tools/test.py is imported from /root/reference through the harness shim (tests/compat: cv2 provider, NumPy aliases, the
reference's pyvotkit extension built into a scratch directory) -- never edited, never copied.  The values the tool
keeps in local variables (best_pscore_id, pscore, penalty, score, lr, crop_box, scale_x ...) are read from the frame
of `siamese_track` when it returns (sys.setprofile), so the tool's own arithmetic -- float32 softmax / exp / sz(),
NumPy promotion to float64 at the window blend -- is what gets recorded.

Frames: windows of data/tennis/*.jpg (real images, the reference's demo clip) stored as uint8.  Weights: the
synthetic_damped checkpoint, loaded through the reference's load_pretrain.  Per frame the fixture keeps the state
before and after, the raw cls/loc the reference network produced, the decode results and (mask variants) the
thresholded full-frame mask + the Refine / mask-column logits."""
import json
import os
import sys
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

from tests.compat import shim  # noqa: E402
from siammask_amd import synth  # noqa: E402

REF = shim.REF
GOLD = os.path.join(REPO, "tests", "golden")
EXP = {"sharp": "siammask_sharp", "base": "siammask_base", "rpn": "siamrpn_resnet"}
CFG = {"sharp": "config_davis.json", "base": "config.json", "rpn": "config.json"}


def call_with_locals(fn, *a, **k):
    """run fn and return (result, its local variables at return)"""
    got = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code is fn.__code__:
            got.update(frame.f_locals)
    sys.setprofile(prof)
    try:
        r = fn(*a, **k)
    finally:
        sys.setprofile(None)
    return r, got


class Recorder(object):
    """transparent proxy around the reference model: records what crosses the Custom.* boundary"""

    def __init__(self, model):
        self._m = model
        self.calls = []

    def __getattr__(self, name):
        v = getattr(self._m, name)
        if name in ("template", "track", "track_mask", "track_refine"):
            def wrapped(*a):
                out = v(*a)
                # snapshot NOW: at B=1 the tool's permute().contiguous().view() aliases the network output and its
                # in-place anchor decode (tools/test.py:209-212) then overwrites it
                snap = tuple(o.clone() for o in out) if isinstance(out, tuple) else (out.clone() if out is not None else None)
                self.calls.append((name, a, snap))
                return out
            return wrapped
        return v


def tennis_frames(ids, window):
    from PIL import Image
    x0, y0, w, h = window
    out = []
    for i in ids:
        rgb = np.asarray(Image.open(os.path.join(REF, "data", "tennis", "%05d.jpg" % i)).convert("RGB"))
        out.append(np.ascontiguousarray(rgb[y0:y0 + h, x0:x0 + w, ::-1]))       # BGR, as cv2.imread gives
    return np.stack(out)


def run_case(variant, frame_ids, window, init_rect, mask, refine):
    shim.install(os.path.join(REF, "experiments", EXP[variant]))
    t = shim.load_tools_test()
    from custom import Custom                                     # the reference's, from experiments/<exp>/custom.py
    from utils.load_helper import load_pretrain
    import tempfile
    cfg = json.load(open(os.path.join(REF, "experiments", EXP[variant], CFG[variant])))
    model = Custom(anchors=cfg["anchors"])
    sd = synth.torch_state_dict(variant, "synthetic_damped")
    with tempfile.NamedTemporaryFile(suffix=".pth", delete=False) as f:
        path = f.name
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, path)
    model = load_pretrain(model, path)
    os.unlink(path)
    model.eval()
    rec = Recorder(model)
    frames = tennis_frames(frame_ids, window)
    x, y, w, h = init_rect
    target_pos = np.array([x + w / 2, y + h / 2])
    target_sz = np.array([w, h])
    hp = cfg.get("hp")
    store = {"frames": frames, "init_rect": np.array(init_rect, dtype=np.float64), "variant": variant,
             "hp_json": json.dumps(hp), "anchors_json": json.dumps(cfg["anchors"]),
             "mask_enable": np.bool_(mask), "refine_enable": np.bool_(refine)}
    with torch.no_grad():
        state = t.siamese_init(frames[0].copy(), target_pos, target_sz, rec, hp, device="cpu")
        store["z_crop"] = rec.calls[-1][1][0][0].numpy().astype(np.uint8)
        assert np.array_equal(store["z_crop"], rec.calls[-1][1][0][0].numpy())       # cv2.resize of uint8 stays integral
        pp = state["p"]                                   # effective hyper-parameters: hp over the TrackerConfig defaults
        store["p_json"] = json.dumps({k: getattr(pp, k) for k in (
            "penalty_k", "window_influence", "lr", "seg_thr", "exemplar_size", "instance_size", "total_stride",
            "out_size", "base_size", "score_size", "context_amount")})
        store["avg_chans"] = np.asarray(state["avg_chans"], dtype=np.float64)
        store["window"] = np.asarray(state["window"], dtype=np.float64)
        store["anchor"] = np.asarray(state["p"].anchor)
        assert store["anchor"].dtype == np.float32
        per = {k: [] for k in ("pos_in", "sz_in", "cls", "loc", "best_id", "pscore_best", "top2_gap", "score_best",
                               "penalty_best", "lr", "pos_out", "sz_out", "delta_yx", "scale_x", "crop_box", "x_crop",
                               "pred_in_crop", "logits", "mask_bits", "polygon", "pscore_dtype")}
        for f in range(1, len(frames)):
            per["pos_in"].append(np.array(state["target_pos"], dtype=np.float64))
            per["sz_in"].append(np.array(state["target_sz"], dtype=np.float64))
            n0 = len(rec.calls)
            state, loc = call_with_locals(t.siamese_track, state, frames[f].copy(), mask, refine, "cpu")
            name, a, out = rec.calls[n0]
            per["x_crop"].append(a[0][0].numpy().astype(np.uint8))
            per["cls"].append(out[0][0].numpy().copy())
            per["loc"].append(out[1][0].numpy().copy())
            ps = loc["pscore"]
            bid = int(loc["best_pscore_id"])
            srt = np.sort(ps)[::-1]
            per["best_id"].append(bid)
            per["pscore_best"].append(float(ps[bid]))
            per["pscore_dtype"].append(str(ps.dtype))
            per["top2_gap"].append(float(srt[0] - srt[1]))
            per["score_best"].append(float(loc["score"][bid]))
            per["penalty_best"].append(float(loc["penalty"][bid]))
            per["lr"].append(float(loc["lr"]))
            per["scale_x"].append(float(loc["scale_x"]))
            per["crop_box"].append(np.array(loc["crop_box"], dtype=np.float64))
            per["pred_in_crop"].append(np.array(loc["pred_in_crop"], dtype=np.float64))
            per["pos_out"].append(np.array(state["target_pos"], dtype=np.float64))
            per["sz_out"].append(np.array(state["target_sz"], dtype=np.float64))
            if mask:
                per["delta_yx"].append(np.array([loc["delta_y"], loc["delta_x"]], dtype=np.int64))
                if refine:
                    logits = rec.calls[n0 + 1][2][0].numpy().reshape(-1).copy()
                else:
                    logits = out[2][0, :, int(loc["delta_y"]), int(loc["delta_x"])].numpy().copy()
                per["logits"].append(logits)
                per["mask_bits"].append(np.packbits(state["mask"] > state["p"].seg_thr))
                per["polygon"].append(np.asarray(state["ploygon"], dtype=np.float64).reshape(4, 2))
    assert set(per["pscore_dtype"]) == {"float64"}, per["pscore_dtype"]      # NumPy >= 2 promotion (tools/test.py:229-236)
    del per["pscore_dtype"]
    for k, v in per.items():
        if v:
            store["f_" + k] = np.stack(v)
    path = os.path.join(GOLD, "tracker_%s.npz" % variant)
    np.savez_compressed(path, **store)
    print("%-6s frames=%d best=%s gap=%s (%.0f KB)" % (variant, len(frames), store["f_best_id"], np.round(store["f_top2_gap"], 5),
                                                     os.path.getsize(path) / 1024.0))
    print("       pos_out", np.round(store["f_pos_out"], 3).tolist())


def main():
    torch.set_num_threads(os.cpu_count())
    os.makedirs(GOLD, exist_ok=True)
    # window of the 854x480 clip around the player; boxes in window coordinates (x, y, w, h)
    run_case("sharp", (0, 1, 2, 3, 4), (230, 60, 320, 240), (118, 38, 92, 158), mask=True, refine=True)
    run_case("base", (10, 11, 12), (230, 60, 320, 240), (150, 80, 44, 66), mask=True, refine=False)
    run_case("rpn", (20, 21, 22), (230, 60, 320, 240), (20, 150, 60, 50), mask=False, refine=False)


if __name__ == "__main__":
    main()
