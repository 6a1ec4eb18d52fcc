"""Device-resident tracker loop for B streams in lock-step: the host logic of tools/test.py
(`siamese_init` :132-170, `siamese_track` :173-311) with every image-sized operation on the MI355X
(crop+resize, network, decode, Refine, mask paste-back).  The host keeps the per-stream scalar state
(target_pos / target_sz update, :241-250 and :302-305); contours / minAreaRect (:285-294) are left to the
caller.  Additive: the reference's tools keep running their own functions through the drop-in Custom.

    tr = DeviceTracker(model, hp={'penalty_k': 0.04, 'window_influence': 0.4, 'lr': 1.0, 'seg_thr': 0.35})
    tr.init(frame_u8_cuda, [(cx, cy), ...], [(w, h), ...])         # siamese_init per stream
    st = tr.track(next_frame_u8_cuda)                              # siamese_track(mask_enable, refine_enable)
    st['target_pos'], st['target_sz'], st['score'], st['mask']     # [B,2], [B,2], [B], uint8 [B,im_h,im_w]
"""
import numpy as np
import torch

from . import preproc


class TrackerConfig(object):
    """utils/tracker_config.py:10-47 (defaults of the reference)"""
    penalty_k = 0.09
    window_influence = 0.39
    lr = 0.38
    seg_thr = 0.3
    exemplar_size = 127
    instance_size = 255
    total_stride = 8
    out_size = 63
    base_size = 8
    context_amount = 0.5

    def __init__(self, hp=None):
        for k, v in (hp or {}).items():
            setattr(self, k, v)
        self.score_size = (self.instance_size - self.exemplar_size) // self.total_stride + 1 + self.base_size


def _mean_colour(frame):
    """np.mean(im, axis=(0, 1)) (tools/test.py:146) on the device, in float64"""
    return frame.to(torch.float64).mean(dim=(0, 1)).cpu().numpy()


class DeviceTracker(object):
    def __init__(self, model, hp=None, pipeline=False):
        """pipeline=True: the frame steps are software-pipelined (Custom.set_pipeline): the host reads the decoded box and updates
        the tracker state (:240-250,302-305) while the Refine mask of the same frame is still being computed on a side stream; the
        mask is joined only where it is pasted back (:257-284) -- same results."""
        self.model = model
        self.pipeline = bool(pipeline)
        self.p = TrackerConfig(hp)
        self.refine = model.variant == "sharp"
        # config_davis.json hp sets out_size 127 for the Refine output; the base head is 63x63
        self.mask_size = int(hp["out_size"]) if hp and "out_size" in hp else (127 if self.refine else 63)
        model.set_tracker_hp(self.p.penalty_k, self.p.window_influence)
        self.state = None

    # -- siamese_init (tools/test.py:132-170) ---------------------------------------------------
    def init(self, frame, target_pos, target_sz):
        p = self.p
        pos = np.asarray(target_pos, dtype=np.float64).reshape(-1, 2)
        sz = np.asarray(target_sz, dtype=np.float64).reshape(-1, 2)
        B = pos.shape[0]
        frames = frame if frame.dim() == 4 else None
        avg = [_mean_colour(frames[b] if frames is not None else frame) for b in range(B)] if frames is not None \
            else [_mean_colour(frame)] * B
        s_z = []
        for b in range(B):
            wc_z = sz[b, 0] + p.context_amount * sz[b].sum()
            hc_z = sz[b, 1] + p.context_amount * sz[b].sum()
            s_z.append(round(np.sqrt(wc_z * hc_z)))
        z = preproc.crop_batch(frame, pos, p.exemplar_size, s_z, avg)
        self.model.template(z)
        if self.pipeline and self.refine:
            self.model.set_pipeline(True)
        H, W = int(frame.shape[-3]), int(frame.shape[-2])
        self.state = {"im_h": H, "im_w": W, "avg_chans": avg, "target_pos": pos.copy(), "target_sz": sz.copy(),
                      "score": np.zeros(B), "mask": None}
        return self.state

    # -- siamese_track (tools/test.py:173-311) ---------------------------------------------------
    def track(self, frame, want_mask=True, keep_crop=False):
        p, st = self.p, self.state
        want_mask = want_mask and self.model.variant != "rpn"         # siamrpn has no mask branch (mask_enable=False)
        pos, sz = st["target_pos"], st["target_sz"]
        B = pos.shape[0]
        s_x = np.empty(B)
        scale_x = np.empty(B)
        crop_box = []
        for b in range(B):
            wc_x = sz[b, 1] + p.context_amount * sz[b].sum()          # (:181-182; w/h swapped as in the reference)
            hc_x = sz[b, 0] + p.context_amount * sz[b].sum()
            s = np.sqrt(wc_x * hc_x)
            scale_x[b] = p.exemplar_size / s
            pad = (p.instance_size - p.exemplar_size) / 2 / scale_x[b]
            s_x[b] = s + 2 * pad
            r = round(s_x[b])
            crop_box.append([pos[b, 0] - r / 2, pos[b, 1] - r / 2, r, r])
        x = preproc.crop_batch(frame, pos, p.instance_size, [round(v) for v in s_x], st["avg_chans"])
        twh = torch.from_numpy(sz * scale_x[:, None]).to(x.device)    # target_sz_in_crop, float64 (:230)
        out = self.model.track_step(x, twh, refine=self.refine and want_mask, mask_head=not self.refine)
        box = out["box"].cpu().numpy()                                # float64: cx, cy, w, h, score, penalty, pscore, best_id
        best = box[:, 7].astype(np.int64)
        ss = p.score_size
        delta_y, delta_x = (best % (ss * ss)) // ss, best % ss        # np.unravel_index (:253-254)
        new_pos, new_sz = pos.copy(), sz.copy()
        for b in range(B):
            pred = box[b, :4] / scale_x[b]                            # pred_in_crop (:240)
            lr = box[b, 5] * box[b, 4] * p.lr                         # penalty * score * lr (:241)
            new_pos[b] = [pred[0] + pos[b, 0], pred[1] + pos[b, 1]]
            new_sz[b] = [sz[b, 0] * (1 - lr) + pred[2] * lr, sz[b, 1] * (1 - lr) + pred[3] * lr]
        masks = None
        if want_mask:
            bbs = [preproc_back_box(crop_box[b], (int(delta_y[b]), int(delta_x[b])), (st["im_w"], st["im_h"]), p,
                                    self.mask_size) for b in range(B)]
            if self.refine:
                self.model.pipeline_join()                            # (a no-op for serial steps)
                logits = out["refine"]
            else:                                                     # base: one column of the 63x63 head (:259-260)
                m = out["mask"]
                idx = torch.arange(B, device=m.device)
                logits = m[idx, :, torch.as_tensor(delta_y, device=m.device), torch.as_tensor(delta_x, device=m.device)]
            masks = preproc.paste_masks(logits, bbs, (st["im_w"], st["im_h"]), seg_thr=p.seg_thr)
        new_pos[:, 0] = np.clip(new_pos[:, 0], 0, st["im_w"])         # (:302-305)
        new_pos[:, 1] = np.clip(new_pos[:, 1], 0, st["im_h"])
        new_sz[:, 0] = np.clip(new_sz[:, 0], 10, st["im_w"])
        new_sz[:, 1] = np.clip(new_sz[:, 1], 10, st["im_h"])
        st.update(target_pos=new_pos, target_sz=new_sz, score=box[:, 4].copy(), mask=masks, best_id=best,
                  delta_yx=np.stack([delta_y, delta_x], 1), crop_box=crop_box, x_crop=x.clone() if keep_crop else None)
        return st


def preproc_back_box(crop_box, delta_yx, im_wh, p, mask_size):
    """tools/test.py:275-279: the box that maps the mask_size x mask_size mask into the image"""
    delta_y, delta_x = delta_yx
    s = crop_box[2] / p.instance_size
    sub_box = [crop_box[0] + (delta_x - p.base_size / 2) * p.total_stride * s,
               crop_box[1] + (delta_y - p.base_size / 2) * p.total_stride * s,
               s * p.exemplar_size, s * p.exemplar_size]
    s = mask_size / sub_box[2]
    return [-sub_box[0] * s, -sub_box[1] * s, im_wh[0] * s, im_wh[1] * s]
