"""Drop-in ``Custom`` modules: the reference's inference surface on top of libsiammask_hip.so.

Mirrors (names, argument meaning, state and error behaviour):
  * experiments/siammask_sharp/custom.py:162-190  -> ``CustomSharp``
  * experiments/siammask_base/custom.py:93-112    -> ``CustomBase``
  * experiments/siamrpn_resnet/custom.py:81-93    -> ``CustomRPN``
  * models/siammask_sharp.py:14-26, models/siamrpn.py:15-23 (attributes the tools read:
    ``anchors``, ``anchor_num``)

What the tools do with it (tools/test.py:559-569,155,201-207,257-261):
    model = Custom(anchors=cfg['anchors']); load_pretrain(model, path); model.eval().to(device)
    model.template(z); cls, loc, mask = model.track_mask(x); m = model.track_refine((dy, dx))
Parameters/buffers carry the reference state-dict names and shapes (siammask_amd/spec.py), so
an official checkpoint loads through the unchanged ``utils/load_helper.load_pretrain``.  The
nn.Module holds the weights only; all arithmetic runs in the HIP library.  There is no CPU
fallback: calling a method with a CPU tensor, or without the built library, raises.
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib, spec


class _Node(nn.Module):
    """Parameter container mirroring one node of the reference module tree."""

    def __init__(self, tree):
        super(_Node, self).__init__()
        for name, sub in tree.items():
            if isinstance(sub, OrderedDict):
                self.add_module(name, _Node(sub))
            else:
                shape, kind = sub
                if kind in ("conv_w", "deconv_w", "bias", "bn_w", "bn_b"):
                    init = torch.ones(shape) if kind == "bn_w" else torch.zeros(shape)
                    self.register_parameter(name, nn.Parameter(init, requires_grad=False))
                elif kind == "bn_mean":
                    self.register_buffer(name, torch.zeros(shape))
                elif kind == "bn_var":
                    self.register_buffer(name, torch.ones(shape))
                elif kind == "bn_nbt":
                    self.register_buffer(name, torch.tensor(0, dtype=torch.long))
                else:
                    raise KeyError(kind)

    def forward(self, *a, **k):
        raise RuntimeError("siammask_amd parameter containers are not callable; use "
                           "Custom.template/track/track_mask/track_refine")


class Custom(nn.Module):
    """Base of the three variants.  Constructor signature follows the reference:
    ``Custom(pretrain=False, anchors=<dict>)``; extra keyword-only knobs:

    dtype      'f32' (default; the reference's precision), 'f16' (fp16 storage, fp32 accumulate) or
               'f16x3' (split-operand fp16: every value of the track path's trunk is an fp16 hi + lo pair,
               a product three MFMA products -- the argmax box index of the fp64 reference on the fp16
               matrix pipe; mask head and Refine in plain fp16).  Env override: SIAMMASK_AMD_DTYPE.
    max_batch  streams tracked in lock-step by this instance (default 1, grows on demand).
    graph      replay captured hipGraphs (default: env SIAMMASK_AMD_GRAPH, else on).
    lazy_mask  sharp only: skip the 3969-channel mask head in track_mask (its result is never
               read when track_refine is used, tools/test.py:256-258) and return None for it.
    pack_cache directory of the packed-weight cache, or None (default: env
               SIAMMASK_AMD_PACK_CACHE, else off).  The BN-folded, MFMA-packed weights are stored as
               <sha256(state dict, dtype, variant, ABI)>.smkpack and uploaded from there on later
               starts instead of folding and repacking the checkpoint again.
    """
    variant = None

    def __init__(self, pretrain=False, anchors=None, o_sz=127, g_sz=127, dtype=None, max_batch=1,
                 graph=None, lazy_mask=False, pack_cache=None, **kwargs):
        super(Custom, self).__init__()
        if anchors is None:
            raise ValueError("Custom(anchors=...) is required (tools/test.py:560)")
        if pretrain:
            raise NotImplementedError("pretrain=True loads 'resnet.model' for training; "
                                      "the MI355X path is inference only")
        self.anchors = anchors
        self.anchor_num = len(anchors["ratios"]) * len(anchors["scales"])
        if self.anchor_num != spec.ANCHOR_NUM:
            raise ValueError("kernels are specialised for 5 anchors (config_*.json); got %d" % self.anchor_num)
        self.o_sz, self.g_sz = o_sz, g_sz
        self.all_anchors = None
        tree = spec.module_tree(self.variant)
        for name, sub in tree.items():
            self.add_module(name, _Node(sub))
        dtype = dtype or os.environ.get("SIAMMASK_AMD_DTYPE", "f32")
        if dtype not in _lib.DTYPE:
            raise ValueError("dtype %r" % (dtype,))
        self._dtype = dtype
        if graph is None:
            graph = os.environ.get("SIAMMASK_AMD_GRAPH", "1") != "0"
        self._graph = bool(graph)
        self._lazy_mask = bool(lazy_mask)
        self._pack_cache = pack_cache if pack_cache is not None else os.environ.get("SIAMMASK_AMD_PACK_CACHE") or None
        self.pack_cache_hit = None          # True / False after the weights were (re)loaded with a cache dir
        self._max_batch = int(max_batch)
        self._ctx = None
        self._ctx_device = None
        self._weights_dirty = True
        self._io = {}
        self._fast = {}
        self._tracked = 0
        self._hp = None
        self._hp_dirty = True
        self.zf = None          # reference attribute (custom.py:174); opaque handle here
        self._replay = {}       # what a failed persistent-sequence launch needs to be re-run (see _guarded)
        self.seq_recovered = 0  # frames re-run on the per-layer kernels after a reported sequence failure

    # -- weights --------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        self._weights_dirty = True
        return super(Custom, self).load_state_dict(state_dict, strict=strict)

    def _apply(self, fn, *a, **k):
        self._weights_dirty = True
        return super(Custom, self)._apply(fn, *a, **k)

    def mark_weights_dirty(self):
        self._weights_dirty = True

    # -- context ----------------------------------------------------------------------------
    def _ensure(self, x, batch, grow=False):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError("siammask_amd runs on the MI355X only: expected a CUDA(HIP) tensor, got %s"
                               % (x.device if isinstance(x, torch.Tensor) else type(x)))
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        L = _lib.lib()
        self._ensure_ctx(dev, batch, grow)
        if self._weights_dirty:
            sd = [(name, np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy()))
                  for name, t in self.state_dict().items() if not name.endswith("num_batches_tracked")]
            path = self._pack_path(sd) if self._pack_cache else None
            loaded = False
            if path and os.path.exists(path):
                try:
                    self._import_packed(path)
                    loaded = True
                except _lib.SmkError:
                    loaded = False          # stale or foreign blob: fall through and rebuild it
            if not loaded:
                for name, a in sd:
                    shape = (ctypes.c_int64 * max(1, a.ndim))(*a.shape)
                    _lib.check(L.smk_set_weight(self._ctx, name.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                                shape, a.ndim))
                _lib.check(L.smk_finalize_weights(self._ctx))
                if path:
                    self.save_packed(path)
            if path:
                self.pack_cache_hit = loaded
            self._weights_dirty = False
            self._tracked = 0
            self.zf = None

    # -- packed-weight cache --------------------------------------------------
    def _pack_path(self, sd):
        import hashlib
        h = hashlib.sha256()
        h.update(("%s|%s|abi%#x" % (self.variant, self._dtype, _lib.lib().smk_version())).encode())
        for name, a in sd:
            h.update(name.encode())
            h.update(str(a.shape).encode())
            h.update(a.tobytes())
        return os.path.join(self._pack_cache, "siammask_%s_%s_%s.smkpack" % (self.variant, self._dtype, h.hexdigest()[:32]))

    def save_packed(self, path):
        """Write the BN-folded, MFMA-packed weights of this (finalized) model to ``path``."""
        if self._ctx is None:
            raise RuntimeError("save_packed(): no device context yet (run template() first)")
        L = _lib.lib()
        n = ctypes.c_uint64(0)
        _lib.check(L.smk_packed_size(self._ctx, ctypes.byref(n)))
        buf = (ctypes.c_ubyte * n.value)()
        _lib.check(L.smk_export_packed(self._ctx, buf, n.value))
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = "%s.tmp%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(bytes(buf))
        os.replace(tmp, path)                # atomic: concurrent ranks may race on the same file

    def _import_packed(self, path):
        with open(path, "rb") as f:
            blob = f.read()
        buf = (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)
        _lib.check(_lib.lib().smk_import_packed(self._ctx, buf, len(blob)))

    def load_packed(self, path, device=None):
        """Upload packed weights written by save_packed(); the module's parameters are NOT updated
        (they are only the checkpoint container) -- use for serving starts without a checkpoint."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._ensure_ctx(dev.index if dev.index is not None else torch.cuda.current_device(), self._max_batch)
        self._import_packed(path)
        self._weights_dirty = False
        self._tracked = 0
        self.zf = None

    def _ensure_ctx(self, dev, batch, grow=True):
        """grow=True (template(), load_packed()): a larger batch or another device re-creates the context -- the cached
        template goes with it, which is fine because a new template follows.  grow=False (track*, decode): the
        context is never silently destroyed under a cached template; a batch beyond max_batch is an error."""
        L = _lib.lib()
        if self._ctx is not None and (dev != self._ctx_device or batch > self._max_batch):
            if not grow:
                if dev != self._ctx_device:
                    raise RuntimeError("input is on cuda:%s but this model's context (weights, cached template) lives on "
                                       "cuda:%s; call template() on the new device first" % (dev, self._ctx_device))
                raise RuntimeError("batch %d exceeds this context's max_batch %d; call template() with the new batch "
                                   "first (or construct with max_batch=%d)" % (batch, self._max_batch, batch))
            self._destroy()
        if self._ctx is None:
            self._max_batch = max(self._max_batch, batch)
            ctx = ctypes.c_void_p()
            _lib.check(L.smk_create(ctypes.byref(ctx), dev, _lib.DTYPE[self._dtype],
                                    _lib.VARIANT[self.variant], self._max_batch))
            self._ctx, self._ctx_device = ctx, dev
            _lib.check(L.smk_set_graph_mode(ctx, 1 if self._graph else 0))
            self._weights_dirty = True
            self._hp_dirty = True
            self._io = {}
            self._fast = {}

    def _destroy(self):
        if self._ctx is not None:
            torch.cuda.synchronize()
            _lib.lib().smk_destroy(self._ctx)
        self._ctx = None
        self._io = {}
        self._fast = {}
        self.zf = None
        self._tracked = 0
        # the ring, the pipeline mode and the replay closures belonged to that context
        self._ring = None
        self._replay = {}
        self._pipeline = False

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _buf(self, key, shape, device, dtype=torch.float32):
        t = self._io.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != device or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._io[key] = t
        return t

    def _stage_in(self, key, x, size):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != size or x.shape[3] != size:
            raise ValueError("expected a [B,3,%d,%d] tensor, got %s" % (size, size, tuple(x.shape)))
        if self._graph:
            # stable pointer => the captured graph is replayed
            buf = self._buf(key, tuple(x.shape), x.device)
            buf.copy_(x)
            return buf
        return x.to(torch.float32).contiguous()

    def _out(self, key, shape, device, dtype=torch.float32):
        # pipelined steps: the library's private side stream writes `refine` / `mask` after track_step returns, and torch's caching
        # allocator does not know that stream -- a fresh tensor dropped by the caller before pipeline_join() could be handed out
        # again while the tail still writes into it.  Persistent buffers (as in graph mode) cannot be recycled under the tail.
        if self._graph or getattr(self, "_pipeline", 0):
            return self._buf(key, shape, device, dtype)
        return torch.empty(shape, dtype=dtype, device=device)

    # -- a persistent-sequence failure is reported by the call that produced the invalid frame -------------------
    def _guarded(self, stage, run):
        """Enqueue one entry point of the reference surface (``run``) and, when it contained a launch of the persistent
        per-XCD sequence kernel, wait for it and read the kernel's failure flag (smk_seq_sync_check: a stream synchronisation
        exactly where the reference's callers synchronise anyway -- tools/test.py:205 `.cpu()` -- and nothing at all for
        batches that do not use the sequence).  On a reported failure the library has already switched this context to the
        per-layer kernels and dropped the cached template; the frame is re-run transparently from the inputs kept in
        ``self._replay`` (template -> track -> refine as far as ``stage`` needs), so the caller never sees an invalid frame
        and never an exception it would have to answer with a re-submit."""
        L = _lib.lib()

        def once():
            run()
            _lib.check(L.smk_seq_sync_check(self._ctx, _lib.current_stream_ptr(), None))
        try:
            return once()
        except _lib.SmkError as e:
            # (raised by the check behind the call, or by the entry point itself when an earlier, unguarded call -- the
            #  asynchronous track_step -- left the flag set)
            if e.code != _lib.E_SEQ:
                raise
            failure = e
        order = ("template", "track", "refine")
        need = order[:order.index(stage)]
        if any(self._replay.get(st) is None for st in need):
            # (e.g. template(); track_step(); track_refine(pos): no track() of THIS frame was recorded -- nothing valid to re-run)
            raise failure
        self.seq_recovered += 1
        for st in need:
            self._replay[st]()
        once()

    # -- the reference surface ------------------------------------------------------------------
    def template(self, template):
        """custom.py:173-174 -- caches the template features (and conv_kernel(zf)) on device."""
        B = template.shape[0]
        self._ensure(template, B, grow=True)
        with torch.cuda.device(self._ctx_device):
            z = self._stage_in("z", template, spec.TEMPLATE_SIZE)
            if not self._graph:
                z = z.clone()        # kept for a re-run (the staged buffer of graph mode is ours already)

            def run():
                _lib.check(_lib.lib().smk_template(self._ctx, z.data_ptr(), B, _lib.current_stream_ptr()))
            self._replay = {"template": run}
            self._guarded("template", run)
        self.zf = ("device-resident", B)
        self._tracked = 0

    def _track(self, search, flags, want_mask):
        if self.zf is None:
            raise RuntimeError("template() must be called before track()/track_mask()")
        B = search.shape[0]
        self._ensure(search, B)
        if self.zf is None:
            raise RuntimeError("weights changed since template(); call template() again")
        dev = search.device
        with torch.cuda.device(self._ctx_device):
            x = self._stage_in("x", search, spec.SEARCH_SIZE)
            cls = self._out("cls", (B, 2 * self.anchor_num, spec.SCORE_SIZE, spec.SCORE_SIZE), dev)
            loc = self._out("loc", (B, 4 * self.anchor_num, spec.SCORE_SIZE, spec.SCORE_SIZE), dev)
            mask = None
            if want_mask:
                mask = self._out("mask", (B, spec.MASK_OUT ** 2, spec.SCORE_SIZE, spec.SCORE_SIZE), dev)
            def run():
                _lib.check(_lib.lib().smk_track(
                    self._ctx, x.data_ptr(), B, flags, cls.data_ptr(), loc.data_ptr(),
                    mask.data_ptr() if mask is not None else None, _lib.current_stream_ptr()))
            self._replay["track"] = run          # (x: the staged buffer in graph mode, else the caller's tensor -- kept alive here)
            self._guarded("track", run)
        if self._graph:
            cls, loc = cls.clone(), loc.clone()      # small; mask stays a view of the I/O buffer
        return cls, loc, mask

    def track(self, search):
        """custom.py:176-179 -> (rpn_pred_cls [B,10,25,25], rpn_pred_loc [B,20,25,25])."""
        cls, loc, _ = self._track(search, _lib.TRACK_BOX, False)
        self._tracked = 0
        return cls, loc

    # -- additive API: on-device decode + fused per-frame step -------------------
    def set_tracker_hp(self, penalty_k=0.04, window_influence=0.4):
        """hp of the host decode (config_*.json 'hp'); anchors come from self.anchors exactly as
        utils/anchors.py:38-50 builds them: ws/hs integer-truncated, or rounded to `round_dight` decimals when
        anchors['round_dight'] > 0; stored float32 like the reference's anchor table (:29)."""
        import math
        self._hp = (float(penalty_k), float(window_influence))
        wh = []
        size = self.anchors["stride"] ** 2
        rd = int(self.anchors.get("round_dight", 0) or 0)
        if int(self.anchors.get("anchor_density", 1) or 1) != 1:
            raise ValueError("anchor_density != 1 is not supported by the device decode (config_*.json use 1)")
        for r in self.anchors["ratios"]:
            if rd > 0:
                ws = round(math.sqrt(size * 1. / r), rd)
                hs = round(ws * r, rd)
            else:
                ws = int(math.sqrt(size * 1. / r))
                hs = int(ws * r)
            for sc in self.anchors["scales"]:
                # the table stores corners (-w/2, -h/2, w/2, h/2) in float32 and the tools take x2-x1, y2-y1
                w2, h2 = np.float32(ws * sc * 0.5), np.float32(hs * sc * 0.5)
                wh += [w2 - (-w2), h2 - (-h2)]
        self._anchor_wh = np.asarray(wh, dtype=np.float32)
        self._hp_dirty = True

    def _push_hp(self):
        if getattr(self, "_hp", None) is None:
            self.set_tracker_hp()
        if self._hp_dirty:
            _lib.check(_lib.lib().smk_set_decode_params(
                self._ctx, self._anchor_wh.ctypes.data_as(ctypes.c_void_p), 5, int(self.anchors["stride"]),
                self._hp[0], self._hp[1]))
            self._hp_dirty = False

    def decode(self, cls, loc, target_wh):
        """Device restatement of tools/test.py:205-254 for B streams.
        target_wh: [B,2] tensor (any float dtype / device; used as float64 like the tool's target_sz*scale_x, :230),
        target size in crop pixels (w,h).
        -> (pos [B,2] int32 (y,x), box [B,8] float64: cx,cy,w,h,score,penalty,pscore,best_id)."""
        B = cls.shape[0]
        if cls.dtype != torch.float32 or loc.dtype != torch.float32:
            raise ValueError("decode() takes the float32 cls/loc the network returns")
        self._ensure(cls, B)
        self._push_hp()
        dev = cls.device
        with torch.cuda.device(self._ctx_device):
            pos = torch.empty((B, 2), dtype=torch.int32, device=dev)
            box = torch.empty((B, 8), dtype=torch.float64, device=dev)
            twh = target_wh.to(dev, torch.float64).contiguous()
            _lib.check(_lib.lib().smk_decode(self._ctx, cls.contiguous().data_ptr(), loc.contiguous().data_ptr(), B,
                                            twh.data_ptr(), pos.data_ptr(), box.data_ptr(), _lib.current_stream_ptr()))
        return pos, box

    def track_step(self, search, target_wh, refine=None, mask_head=True, stage=True):
        """One frame for B streams without leaving the device: track(_mask) -> decode -> refine at
        the decoded positions, replayed as ONE captured graph.
        -> dict(cls, loc, mask, box [B,8], refine [B,16129] or None).  With graph replay the
        returned tensors are views of persistent I/O buffers (valid until the next call).
        ``box`` is float64 [B,8] (cx, cy, w, h, score, penalty, pscore, best_id), target_wh float64 (:230).
        stage=False: ``search`` (float32) / ``target_wh`` (float64) are caller-owned persistent CUDA buffers
        (e.g. a ring of pre-staged crops); they are read in place (no staging copy) and the
        captured graph is keyed on their addresses."""
        if self.zf is None:
            raise RuntimeError("template() must be called before track_step()")
        B = search.shape[0]
        if refine is None:
            refine = self.variant == "sharp"
        if getattr(self, "_ring", None) is not None and B != self._ring[2]:
            raise ValueError("the result ring was set for batch %d, got %d" % (self._ring[2], B))
        if not stage:
            # serving fast path: everything about this call is cached per (input buffers, options)
            key = (search.data_ptr(), target_wh.data_ptr(), B, bool(refine), bool(mask_head))
            hit = self._fast.get(key) if self._ctx is not None and not self._weights_dirty and not self._hp_dirty else None
            if hit is not None:
                self._fast[key] = self._fast.pop(key)    # most recently used last
                args, out = hit
                _lib.check(self._smk_step(self._ctx, *args, _lib.current_stream_ptr()))
                if "track" in self._replay:
                    del self._replay["track"]
                return out
        self._ensure(search, B)
        self._push_hp()
        flags = _lib.TRACK_BOX if self.variant == "rpn" else _lib.TRACK_MASK
        want_mask = self.variant != "rpn" and mask_head
        if self.variant != "rpn" and not mask_head:
            flags |= _lib.TRACK_NO_MASK_HEAD
        dev = search.device
        S, A = spec.SCORE_SIZE, self.anchor_num
        with torch.cuda.device(self._ctx_device):
            if stage:
                x = self._stage_in("x", search, spec.SEARCH_SIZE)
                twh = self._buf("twh", (B, 2), dev, torch.float64)
                twh.copy_(target_wh)
            else:
                if (search.dtype != torch.float32 or not search.is_contiguous() or tuple(search.shape[1:]) !=
                        (3, spec.SEARCH_SIZE, spec.SEARCH_SIZE)):
                    raise ValueError("stage=False needs a contiguous float32 [B,3,255,255] tensor")
                if target_wh.dtype != torch.float64 or not target_wh.is_contiguous() or not target_wh.is_cuda:
                    raise ValueError("stage=False needs a contiguous float64 CUDA target_wh [B,2]")
                x, twh = search, target_wh
            cls = self._out("cls", (B, 2 * A, S, S), dev)
            loc = self._out("loc", (B, 4 * A, S, S), dev)
            mask = self._out("mask", (B, spec.MASK_OUT ** 2, S, S), dev) if want_mask else None
            box = self._out("box", (B, 8), dev, torch.float64)
            ref = self._out("refine", (B, spec.REFINE_OUT ** 2), dev) if refine else None
            args = (x.data_ptr(), B, flags, twh.data_ptr(), cls.data_ptr(), loc.data_ptr(),
                    mask.data_ptr() if mask is not None else None, box.data_ptr(),
                    ref.data_ptr() if ref is not None else None)
            self._smk_step = _lib.lib().smk_step
            _lib.check(self._smk_step(self._ctx, *args, _lib.current_stream_ptr()))
        self._tracked = B if self.variant != "rpn" else 0
        self._replay.pop("track", None)      # the recorded track() is not the frame the context holds any more (see _guarded)
        out = {"cls": cls, "loc": loc, "mask": mask, "box": box, "refine": ref}
        if not stage and self._graph:
            self._fast.pop(key, None)
            while len(self._fast) >= 32:                 # least recently used out (dicts keep insertion order)
                self._fast.pop(next(iter(self._fast)))
            self._fast[key] = (args, out)
        return out

    def set_result_ring(self, rows, batch=None, refine=True):
        """Keep the last ``rows`` frames' results on the device (smk_set_result_ring): every track_step then also writes its
        decoded box and its fp16 Refine logits into row (frames % rows) of the returned tensors
        (box_ring [rows,B,8] float64, refine_ring [rows,B,16129] float16 or None) -- the rows an end-of-batch gather sends
        (siammask_amd.dist.ResultGather), with no per-frame copy by the caller.  rows = 0 switches it off."""
        if self._ctx is None:
            raise RuntimeError("set_result_ring(): run template() first")
        B = int(batch or (self.zf[1] if self.zf else self._max_batch))
        dev = torch.device("cuda", self._ctx_device)
        self._fast = {}
        if rows <= 0:
            _lib.check(_lib.lib().smk_set_result_ring(self._ctx, None, None, 0, 0))
            self._ring = None
            return None, None
        with torch.cuda.device(self._ctx_device):
            box = torch.zeros((rows, B, 8), dtype=torch.float64, device=dev)
            ref = torch.zeros((rows, B, spec.REFINE_OUT ** 2), dtype=torch.float16, device=dev) if refine else None
        _lib.check(_lib.lib().smk_set_result_ring(self._ctx, box.data_ptr(), ref.data_ptr() if ref is not None else None, rows, B))
        self._ring = (box, ref, B)
        return box, ref

    def set_pipeline(self, on=True):
        """Software-pipeline track_step (smk_set_pipeline): the Refine / mask tail of frame f runs on a side stream beside the
        stem + layer1 launches of frame f + 1 -- the next frame's crop only needs the decoded box (tools/test.py:240-250,302-308),
        the mask is an output (:257-284).  ``box`` / ``cls`` / ``loc`` of a track_step are complete in stream order as before;
        ``refine`` / ``mask`` (and the ring's logits row) of frame f are complete behind ``pipeline_join()`` or once the next
        track_step's box is.  Bit-identical to the serial step."""
        if self._ctx is None:
            raise RuntimeError("set_pipeline(): run template() first")
        depth = int(on)           # True = 1; 2 = throughput mode: the Refine chain + mask head of frame f beside the heads of frame f + 1
        _lib.check(_lib.lib().smk_set_pipeline(self._ctx, depth))
        self._pipeline = depth

    def pipeline_join(self, stream=None, launch_pending=True):
        """Order ``stream`` (default: the current stream) behind the outstanding Refine / mask tail of the last pipelined
        track_step; a no-op when there is none.  launch_pending=False (depth 2): only behind what has been launched so far --
        the second part of the last frame's tail keeps waiting for the next track_step (smk_pipeline_observe)."""
        if self._ctx is None:
            return
        sp = _lib.current_stream_ptr() if stream is None else ctypes.c_void_p(stream.cuda_stream)
        fn = _lib.lib().smk_pipeline_join if launch_pending else _lib.lib().smk_pipeline_observe
        _lib.check(fn(self._ctx, sp))

    def result_ring_frames(self, reset=False):
        """frames committed to the ring so far (synchronises the current stream)"""
        n = ctypes.c_int(0)
        _lib.check(_lib.lib().smk_result_ring_cursor(self._ctx, ctypes.byref(n), 1 if reset else 0, _lib.current_stream_ptr()))
        return n.value

    def seq_status(self):
        """(workgroups per persistent sequence launch or 0, device error flag); raises if the kernel reported an error"""
        if self._ctx is None:
            return 0, 0
        g, e = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().smk_seq_status(self._ctx, ctypes.byref(g), ctypes.byref(e)))
        return g.value, e.value

    # -- per-launch profiling (HIP events around every kernel; bypasses graph replay) -----------
    def profile(self, enable=True):
        """enable: False / 0 off; True / 1 per-LAYER attribution (merged launches are split into their members);
        2 = per-LAUNCH attribution (the launch structure of the timed path, merged launches kept)"""
        if self._ctx is None:
            raise RuntimeError("profile(): run template() first")
        _lib.check(_lib.lib().smk_profile(self._ctx, int(enable)))

    def profile_dump(self):
        """-> list of {'id','kernel','calls','ms','flop','bytes'} (algorithmic work), then resets."""
        import json
        buf = ctypes.create_string_buffer(1 << 18)
        _lib.check(_lib.lib().smk_profile_dump(self._ctx, buf, len(buf)))
        return json.loads(buf.value.decode())

    def __repr__(self):
        return "%s(variant=%s, dtype=%s, graph=%s)" % (type(self).__name__, self.variant, self._dtype, self._graph)


class CustomRPN(Custom):
    """experiments/siamrpn_resnet/custom.py:81-93 (box only)."""
    variant = "rpn"


class CustomBase(Custom):
    """experiments/siammask_base/custom.py:93-112 (3-branch, 63x63 mask head, no Refine)."""
    variant = "base"

    def track_mask(self, search):
        """-> (cls, loc, pred_mask [B,3969,25,25])."""
        out = self._track(search, _lib.TRACK_MASK, True)
        self._tracked = search.shape[0]
        return out


class CustomSharp(CustomBase):
    """experiments/siammask_sharp/custom.py:162-190 (mask branch + Refine)."""
    variant = "sharp"

    def track_mask(self, search):
        if self._lazy_mask:
            cls, loc, _ = self._track(search, _lib.TRACK_MASK | _lib.TRACK_NO_MASK_HEAD, False)
            self._tracked = search.shape[0]
            return cls, loc, None
        return super(CustomSharp, self).track_mask(search)

    def track_refine(self, pos):
        """custom.py:188-190.  ``pos`` = (y, x) for the whole batch (reference semantics), or a
        [B,2] int array / tensor of per-stream positions (extension for batched streams).
        -> [B, 127*127] mask logits."""
        if not self._tracked:
            raise RuntimeError("track_refine() requires a preceding track_mask()")
        B = self._tracked
        L = _lib.lib()
        dev = torch.device("cuda", self._ctx_device)
        with torch.cuda.device(self._ctx_device):
            out = self._out("refine", (B, spec.REFINE_OUT ** 2), dev)
            if isinstance(pos, torch.Tensor) and pos.is_cuda:
                p = pos.to(torch.int32).contiguous().view(-1)
                if p.numel() != 2 * B:
                    raise ValueError("pos tensor must have shape [B,2]")
                self._guarded("refine", lambda: _lib.check(
                    L.smk_refine(self._ctx, p.data_ptr(), 1, B, out.data_ptr(), _lib.current_stream_ptr())))
            else:
                a = np.asarray(pos.cpu() if isinstance(pos, torch.Tensor) else pos, dtype=np.int32)
                if a.ndim == 1:
                    a = np.tile(a.reshape(1, 2), (B, 1))
                if a.shape != (B, 2):
                    raise ValueError("pos must be (y, x) or [B,2]")
                a = np.ascontiguousarray(a)
                self._guarded("refine", lambda: _lib.check(
                    L.smk_refine(self._ctx, a.ctypes.data_as(ctypes.c_void_p), 0, B, out.data_ptr(), _lib.current_stream_ptr())))
        return out

    def debug_tensor(self, name):
        """Read an internal activation back as f32 NCHW (parity tests)."""
        L = _lib.lib()
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(L.smk_debug_read(self._ctx, name.encode(), None, ctypes.byref(c), ctypes.byref(h),
                                    ctypes.byref(w), None))
        B = self._tracked or (self.zf[1] if self.zf else 1)
        with torch.cuda.device(self._ctx_device):
            t = torch.empty((B, c.value, h.value, w.value), dtype=torch.float32,
                            device=torch.device("cuda", self._ctx_device))
            _lib.check(L.smk_debug_read(self._ctx, name.encode(), t.data_ptr(), ctypes.byref(c), ctypes.byref(h),
                                        ctypes.byref(w), _lib.current_stream_ptr()))
        return t


CustomBase.debug_tensor = CustomSharp.debug_tensor
CustomRPN.debug_tensor = CustomSharp.debug_tensor

VARIANT_CLASS = {"rpn": CustomRPN, "base": CustomBase, "sharp": CustomSharp}


def build(variant="sharp", anchors=None, **kw):
    anchors = anchors or {"stride": 8, "ratios": [0.33, 0.5, 1, 2, 3], "scales": [8], "round_dight": 0}
    return VARIANT_CLASS[variant](anchors=anchors, **kw)
