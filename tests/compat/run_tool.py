#!/usr/bin/env python
"""Run one of the reference's UNCHANGED tools under the harness shim, in its own process, and report what happened as
one JSON line on stdout (used by tests/test_dropin_tools.py).

    run_tool.py demo  <custom_dir> <n_frames>     # tools/demo.py, whole file, as __main__ (runpy)
    run_tool.py trace <custom_dir> <n_frames> <out.npz>   # the same, with tools.test.siamese_track observed: per-frame
                                                  # target_pos / target_sz / score / thresholded mask -> out.npz, wall time per frame
    run_tool.py main  <custom_dir>                # the model set-up statements of tools/test.py main() (:556-569)
                                                  # followed by siamese_init / siamese_track on two tennis frames

<custom_dir> is what the reference's test*.sh put on PYTHONPATH ahead of the repo root: the experiment directory of
the reference (control run) or dropin/<variant> (the MI355X path)."""
import json
import os
import runpy
import sys
import tempfile
import traceback
import warnings

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.compat import shim  # noqa: E402
from siammask_amd import synth  # noqa: E402


def write_checkpoint(variant):
    """the synthetic checkpoint in the layout the official .pth files have ('state_dict', 'module.' prefixes)"""
    sd = synth.torch_state_dict(variant, "synthetic_damped")
    f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
    f.close()
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, f.name)
    return f.name


def variant_of(custom_dir):
    d = os.path.basename(os.path.normpath(custom_dir))
    return {"siammask_sharp": "sharp", "siammask_base": "base", "siamrpn_resnet": "rpn"}.get(d, d)


def main():
    mode, custom_dir = sys.argv[1], sys.argv[2]
    variant = variant_of(custom_dir)
    exp = {"sharp": "siammask_sharp", "base": "siammask_base", "rpn": "siamrpn_resnet"}[variant]
    cfg_path = os.path.join(shim.REF, "experiments", exp, "config_davis.json" if variant == "sharp" else "config.json")
    info = shim.install(custom_dir)
    from tests.compat import cv2_stub
    ckpt = write_checkpoint(variant)
    out = {"mode": mode, "variant": variant, "region_ext": info["region"], "error": None}
    try:
        if mode in ("demo", "trace"):
            n = int(sys.argv[3])
            trace_out = sys.argv[4] if mode == "trace" else None      # (sys.argv is replaced by demo.py's own below)
            rec = []
            if mode == "trace":
                # observe the unchanged tool: demo.py binds `from tools.test import *` to what the module holds at that moment
                import time as _time
                t = shim.load_tools_test()
                orig_track = t.siamese_track

                def observed(state, im, *a, **k):
                    t0 = _time.perf_counter()
                    st = orig_track(state, im, *a, **k)
                    if torch.cuda.is_available():
                        torch.cuda.synchronize()
                    rec.append(dict(pos=np.array(st["target_pos"], dtype=np.float64), sz=np.array(st["target_sz"], dtype=np.float64),
                                    score=float(st["score"]), mask=np.packbits(np.asarray(st["mask"]) > st["p"].seg_thr),
                                    mask_shape=np.asarray(st["mask"]).shape, sec=_time.perf_counter() - t0))
                    return st
                t.siamese_track = observed
            frames = tempfile.mkdtemp(prefix="smk_demo_frames_")
            for i in range(n):
                os.symlink(os.path.join(shim.REF, "data", "tennis", "%05d.jpg" % i), os.path.join(frames, "%05d.jpg" % i))
            sys.argv = ["demo.py", "--resume", ckpt, "--config", cfg_path, "--base_path", frames] + (
                [] if torch.cuda.is_available() else ["--cpu"])
            try:
                g = runpy.run_path(shim.ref_file("tools", "demo.py"), run_name="__main__")
            finally:
                mod = sys.modules.get("custom")
                out["custom_file"] = getattr(mod, "__file__", None)
                out["custom_class_module"] = getattr(getattr(mod, "Custom", None), "__module__", None)
            st = g["state"]
            if mode == "trace":
                np.savez_compressed(trace_out, pos=np.stack([r["pos"] for r in rec]), sz=np.stack([r["sz"] for r in rec]),
                                    score=np.array([r["score"] for r in rec]), mask=np.stack([r["mask"] for r in rec]),
                                    mask_shape=np.array(rec[0]["mask_shape"]), sec=np.array([r["sec"] for r in rec]))
                out["sec_per_frame_median"] = float(np.median([r["sec"] for r in rec]))
            out.update(frames=int(g["f"]) + 1, target_pos=[float(v) for v in st["target_pos"]],
                       target_sz=[float(v) for v in st["target_sz"]], score=float(st["score"]),
                       mask_shape=list(np.asarray(st["mask"]).shape), cv2_calls=dict(cv2_stub.CALLS))
        else:
            t = shim.load_tools_test()
            cfg = json.load(open(cfg_path))
            # ---- tools/test.py:556-569, statement for statement --------------------------------------------
            from custom import Custom
            model = Custom(anchors=cfg['anchors'])
            model = t.load_pretrain(model, ckpt)
            model.eval()
            device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
            model = model.to(device)
            # ---------------------------------------------------------------------------------------------
            out["custom_file"] = sys.modules["custom"].__file__
            out["custom_class_module"] = type(model).__module__
            out["n_state_dict"] = len(model.state_dict())
            ck = {k[len("module."):] for k in torch.load(ckpt)["state_dict"]}
            have = set(model.state_dict())
            out["ckpt_keys_not_in_model"] = sorted(ck - have)
            out["model_keys_not_in_ckpt"] = sorted(k for k in have - ck if not k.endswith("num_batches_tracked"))
            out["anchors_attr"] = model.anchors == cfg["anchors"] and model.anchor_num == 5
            ims = [cv2_stub.imread(os.path.join(shim.REF, "data", "tennis", "%05d.jpg" % i)) for i in range(2)]
            x, y, w, h = cv2_stub.SELECT_ROI
            state = t.siamese_init(ims[0], np.array([x + w / 2, y + h / 2]), np.array([w, h]), model, cfg.get("hp"),
                                   device=device)
            state = t.siamese_track(state, ims[1], variant != "rpn", variant == "sharp", device)
            out.update(target_pos=[float(v) for v in state["target_pos"]], score=float(state["score"]))
    except BaseException as e:  # noqa: BLE001 -- reported, the caller decides whether it is the expected stop
        tb = traceback.extract_tb(e.__traceback__)
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        out["error_where"] = ["%s:%d %s" % (os.path.relpath(f.filename, "/"), f.lineno, f.name) for f in tb[-6:]]
    finally:
        os.unlink(ckpt)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
