"""CPU tests of the host logic of libsiammask_hip.so (no GPU needed):
the C-ABI library loads and exports every declared symbol, and the weight packing order +
the kernels' own row/tap decode + gather-offset functions (shared host/device code in
siammask_amd/csrc/smk_kernels.h), walked on the host by smk_host_conv2d_ex, reproduce the
oracle's conv2d for every geometry class on the path, including the Refine windows
(custom.py:133-135), the template centre crop (custom.py:21-24) and nearest upsampling
(custom.py:150-152)."""
import ctypes
import re
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from siammask_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    declared = set()
    for h in ("siammask_hip.h", "siammask_hip_test.h"):         # product ABI + test / measurement entry points
        hdr = open(os.path.join(REPO, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)        # (prose in comments mentions entry points of the other header)
        declared |= set(re.findall(r"\b(smk_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("smk_ctx")
    product = set(re.findall(r"\b(smk_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "siammask_hip.h")).read(), flags=re.S)))
    assert not {"smk_tune", "smk_profile", "smk_bench_conv", "smk_op_conv2d_ex", "smk_debug_read"} & product    # the product header stays free of them
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.smk_version() >> 16 == 1


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    ctx = ctypes.c_void_p()
    rc = L.smk_create(ctypes.byref(ctx), 0, 0, 2, 1)
    assert rc == -5 and b"no HIP device" in L.smk_last_error()
    with pytest.raises(_lib.SmkError):
        _lib.check(rc)


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def host_conv(x, w, b=None, res=None, pos=None, **kw):
    g = _lib.ConvGeom()
    B, Cin, H, W = x.shape
    g.B, g.Cin, g.H, g.W = B, Cin, H, W
    g.Cout, g.k = w.shape[0], w.shape[2]
    g.stride, g.pad, g.dil = kw.get("stride", 1), kw.get("pad", 0), kw.get("dil", 1)
    for f in ("relu", "res_mode", "win", "ups", "Hl", "Wl", "org_y", "org_x", "pos_mul", "pos_add",
              "cin_off", "cin_len"):
        setattr(g, f, kw.get(f, 0))
    Hl = g.Hl if (g.win or g.ups) else H
    Wl = g.Wl if (g.win or g.ups) else W
    Ho = (Hl + 2 * g.pad - g.dil * (g.k - 1) - 1) // g.stride + 1
    Wo = (Wl + 2 * g.pad - g.dil * (g.k - 1) - 1) // g.stride + 1
    y = np.zeros((B, g.Cout, Ho, Wo), dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    res = None if res is None else np.ascontiguousarray(res, dtype=np.float32)
    pos = None if pos is None else np.ascontiguousarray(pos, dtype=np.int32)
    _lib.check(_lib.lib().smk_host_conv2d_ex(ctypes.byref(g), _fp(x), _fp(w), _fp(b), _fp(res), _fp(pos), _fp(y)))
    return y


RNG = np.random.default_rng(7)


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,hw", [
    (3, 16, 7, 2, 0, 1, 31),      # stem class: 7x7 s2 p0, Cin=3 padded to 8
    (16, 24, 1, 1, 0, 1, 9),      # 1x1
    (16, 16, 3, 1, 1, 1, 9),      # 3x3 s1 p1
    (16, 8, 3, 2, 0, 1, 13),      # 3x3 s2 p0 (layer2.0)
    (8, 8, 3, 1, 2, 2, 11),       # 3x3 d2 p2 (layer3.1-5)
    (8, 12, 3, 1, 0, 1, 9),       # 3x3 p0 (conv_search / conv_kernel)
    (4, 1, 3, 1, 1, 1, 10),       # Refine tail: Cin=4 (padded to 8), Cout=1
])
def test_host_walk_matches_oracle_conv(cin, cout, k, stride, pad, dil, hw):
    x = RNG.normal(size=(2, cin, hw, hw)).astype(np.float32)
    w = RNG.normal(size=(cout, cin, k, k)).astype(np.float32)
    b = RNG.normal(size=(cout,)).astype(np.float32)
    ref = O.conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), stride, pad, dil)
    got = host_conv(x, w, b, stride=stride, pad=pad, dil=dil)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-4
    res = RNG.normal(size=ref.shape).astype(np.float32)
    got = host_conv(x, w, b, res=res, stride=stride, pad=pad, dil=dil, relu=1, res_mode=1)
    assert np.abs(got - np.maximum(ref + res, 0)).max() < 1e-4
    got = host_conv(x, w, b, res=res, stride=stride, pad=pad, dil=dil, relu=1, res_mode=2)
    assert np.abs(got - (np.maximum(ref, 0) + res)).max() < 1e-4


def test_refine_window_and_positions():
    """F.pad(f, n)[..., m*y:m*y+S, m*x:m*x+S] then conv3x3 p1 (custom.py:133-135)."""
    w = RNG.normal(size=(16, 8, 3, 3)).astype(np.float32)
    pos = np.array([[0, 24], [12, 12], [24, 3]], dtype=np.int32)
    for mul, padn, S, F in ((1, 4, 15, 31), (2, 8, 31, 63)):
        f = RNG.normal(size=(3, 8, F, F)).astype(np.float32)
        fp = np.pad(f, ((0, 0), (0, 0), (padn, padn), (padn, padn))).astype(np.float64)
        ref = np.concatenate([
            O.conv2d(fp[b:b + 1, :, mul * y:mul * y + S, mul * x:mul * x + S], w.astype(np.float64), None, 1, 1, 1)
            for b, (y, x) in enumerate(pos)])
        got = host_conv(f, w, pos=pos, pad=1, win=1, Hl=S, Wl=S, pos_mul=mul, pos_add=-padn)
        assert np.abs(got - ref).max() < 1e-4


def test_template_centre_crop_and_gather():
    """ResDownS crop x[:, :, 4:-4, 4:-4] folded into the 1x1 conv (custom.py:21-24) and the
    corr_feature[:, :, y, x] gather of Refine (custom.py:145) as a 1x1 window."""
    x = RNG.normal(size=(2, 16, 15, 15)).astype(np.float32)
    w = RNG.normal(size=(8, 16, 1, 1)).astype(np.float32)
    ref = O.conv2d(x.astype(np.float64), w.astype(np.float64))[:, :, 4:-4, 4:-4]
    got = host_conv(x, w, win=1, Hl=7, Wl=7, org_y=4, org_x=4)
    assert np.abs(got - ref).max() < 1e-4
    pos = np.array([[3, 9], [14, 0]], dtype=np.int32)
    ref = np.stack([O.conv2d(x[b:b + 1, :, y:y + 1, xx:xx + 1].astype(np.float64), w.astype(np.float64))[0]
                    for b, (y, xx) in enumerate(pos)])
    got = host_conv(x, w, pos=pos, win=1, Hl=1, Wl=1, pos_mul=1)
    assert np.abs(got - ref).max() < 1e-4


def test_nearest_upsample_folded_into_conv():
    x = RNG.normal(size=(2, 8, 15, 15)).astype(np.float32)
    w = RNG.normal(size=(4, 8, 3, 3)).astype(np.float32)
    for size in (31, 61):
        ref = O.conv2d(O.upsample_nearest(x.astype(np.float64), size), w.astype(np.float64), None, 1, 1, 1)
        got = host_conv(x, w, pad=1, ups=1, Hl=size, Wl=size)
        assert np.abs(got - ref).max() < 1e-4


def test_channel_slice():
    x = RNG.normal(size=(1, 24, 6, 6)).astype(np.float32)
    w = RNG.normal(size=(5, 8, 1, 1)).astype(np.float32)
    ref = O.conv2d(x[:, 8:16].astype(np.float64), w.astype(np.float64))
    got = host_conv(x, w, cin_off=8, cin_len=8)
    assert np.abs(got - ref).max() < 1e-4


def test_upsample_map_equals_torch():
    """F.upsample(mode='nearest') index map == (dst*in)//out for the three sizes used."""
    import torch
    import torch.nn.functional as F
    for hin, hout in ((15, 31), (31, 61), (61, 127)):
        t = torch.arange(hin, dtype=torch.float32).view(1, 1, hin, 1).expand(1, 1, hin, hin).contiguous()
        up = F.interpolate(t, size=(hout, hout), mode="nearest")[0, 0, :, 0].numpy().astype(int)
        assert (up == (np.arange(hout) * hin) // hout).all()


def test_pack_cache_key_follows_the_checkpoint(tmp_path):
    """the packed-weight cache is keyed on (state dict, dtype, variant, ABI)."""
    import numpy as np
    from siammask_amd.custom import build
    a = build("rpn", dtype="f16", pack_cache=str(tmp_path))
    sd = [("w", np.arange(6, dtype=np.float32).reshape(2, 3)), ("b", np.zeros(2, dtype=np.float32))]
    p0 = a._pack_path(sd)
    assert p0 == a._pack_path([(n, v.copy()) for n, v in sd])
    sd2 = [("w", sd[0][1] + 1e-3), sd[1]]
    assert a._pack_path(sd2) != p0
    b = build("rpn", dtype="f32", pack_cache=str(tmp_path))
    assert b._pack_path(sd) != p0
    c = build("base", dtype="f16", pack_cache=str(tmp_path))
    assert c._pack_path(sd) != p0


def test_tuning_knobs_read_back():
    """smk_tune / smk_tune_get need no GPU: a knob reads back what was set, unknown names and out-of-range values are errors,
    and the defaults are the measured ones (four producer waves, the layer rule fitted with them, 128-row sequence tiles)."""
    from siammask_amd import _lib
    assert (_lib.tune_get("npw"), _lib.tune_get("wreg_policy"), _lib.tune_get("seq_tall"), _lib.tune_get("a_stage")) == (4, 1, 2, 0)
    assert _lib.tune_get("seq_fuse") == 1 and _lib.tune_get("seq_fused_last") == 0      # conv3 + next conv1 pairs fused; nothing launched
    old = _lib.tune_get("npw")
    try:
        _lib.tune(npw=2)
        assert _lib.tune_get("npw") == 2
        with pytest.raises(RuntimeError):
            _lib.tune(npw=3)
        assert _lib.tune_get("npw") == 2
    finally:
        _lib.tune(npw=old)
    with pytest.raises(RuntimeError):
        _lib.tune_get("no_such_knob")
    with pytest.raises(RuntimeError):
        _lib.tune(no_such_knob=1)


def _plan(B, cin, hw, cout, k, stride=1, pad=0, dil=1, res=False, win=None, dtype="f16"):
    g = _lib.ConvGeom()
    g.B, g.Cin, g.H, g.W = B, cin, hw, hw
    g.Cout, g.k, g.stride, g.pad, g.dil = cout, k, stride, pad, dil
    g.relu = 1
    if win is not None:
        g.win, g.Hl, g.Wl = 1, win, win
    out = [ctypes.c_int(0) for _ in range(4)]
    _lib.check(_lib.lib().smk_host_plan_conv(ctypes.byref(g), _lib.DTYPE[dtype], int(res), *[ctypes.byref(o) for o in out]))
    kernel, bm, bn, cfg = [o.value for o in out]
    return ("igemm", "halo", "wreg", "pp")[kernel], (bm, bn), cfg


def test_layer_rules_are_the_measured_ones():
    """The kernel / workgroup-shape choice per layer is fitted to measurements (profiles/r02_producer_waves_2_vs_4.txt,
    r02_producer_waves_layers_b1_b64.json, r02b_seq_layer_clocks.txt).  smk_host_plan_conv exposes it without a GPU, so a
    change of the rule shows up here instead of as a silent slowdown.  (kernel, workgroup shape, tile code inside a
    persistent sequence or -1)."""
    # B = 8 (headline): layer2 / layer3 / adjust run inside the sequences -> the sequence tile code is what counts
    assert _plan(8, 1024, 31, 256, 1)[2] == 1                                  # l3.c1   64x128
    assert _plan(8, 256, 31, 256, 3, pad=2, dil=2)[2] == 24                    # l3.c2   128 pixels (4 whole rows) x 64, patch shared by the taps
    assert _plan(8, 256, 31, 256, 3, pad=1)[2] == 24                           # l3.0.c2 (dilation 1)
    assert _plan(8, 256, 15, 256, 3, pad=2, dil=2)[2] == 25                    # ... on the 15x15 template: 64 pixels (4 rows) x 64
    assert _plan(8, 256, 31, 1024, 1, res=True)[2] == 3                        # l3.c3   128x256 (two 64-row rounds otherwise)
    assert _plan(8, 512, 31, 1024, 3, pad=1)[2] == 3                           # l3.0.ds 128x256 (long K too, seq_tall = 2)
    assert _plan(8, 256, 63, 512, 3, stride=2)[2] == 0                         # l2.0.ds 64x256 (one round)
    assert _plan(8, 256, 63, 128, 1)[2] == 4                                   # l2.0.c1 on the 63x63 input: 128x128
    assert _plan(8, 128, 31, 128, 3, pad=1)[2] == 25                           # l2.c2   64 pixels (2 whole rows) x 64, patch shared
    assert _plan(8, 128, 31, 512, 1, res=True)[2] == 0                         # l2.c3   64x256
    # B = 8 per launch
    assert _plan(8, 256, 31, 768, 3)[:2] == ("wreg", (96, 256))                # conv_search (N-fused 768): 213 tiles of 96 rows = ONE round (128 rows: 159 tiles
                                                                               # on 256 CUs, each a third longer; round 5, profiles/r05d_wreg96_ab.txt)
    assert _plan(8, 64, 63, 64, 3, pad=1)[0] == "halo"                         # l1.c2: short-K 3x3 stays on the patch kernel
    assert _plan(8, 64, 63, 256, 1, res=True)[:2] == ("igemm", (128, 128))     # l1.c3: large M, short K
    assert _plan(8, 256, 63, 64, 1)[0] == "igemm"                              # l1.c1
    assert _plan(8, 3, 255, 64, 7, stride=2)[0] == "igemm"                     # stem
    assert _plan(8, 512, 31, 128, 3, pad=1, win=15)[:2] == ("wreg", (32, 64))  # Refine v2.0 on its own: 58 tiles of 64 rows -> 114 of 32 (round 6, wreg32)
    # B = 1: almost everything on the register-fed kernel, 64x64 tiles -- 32x64 (round 6, profiles/r06w_wreg_32_row_tiles.txt) where fewer than 140 of those exist
    for args, tile in (((1024, 31, 256, 1), (32, 64)), ((256, 31, 1024, 1), (64, 64)), ((512, 31, 128, 1), (32, 64)), ((256, 63, 64, 1), (32, 64))):
        assert _plan(1, *args)[:2] == ("wreg", tile), args
    assert _plan(1, 256, 31, 256, 3, pad=2, dil=2)[:2] == ("wreg", (32, 64))   # l3.c2: 64 tiles of 64 rows
    assert _plan(2, 256, 31, 256, 3, pad=2, dil=2)[:2] == ("wreg", (32, 64))   # ... 124 at B = 2
    assert _plan(3, 256, 31, 256, 3, pad=2, dil=2)[:2] == ("wreg", (64, 64))   # ... 184 at B = 3
    assert _plan(1, 128, 31, 128, 3, pad=1)[0] == "halo"                       # l2.c2 (K = 1152)
    # B = 64: wide / long-K layers on 128x256 register-fed tiles, narrow short-K ones on LDS-staged 128-row tiles
    assert _plan(64, 1024, 31, 256, 1)[:2] == ("wreg", (128, 256))             # l3.c1
    # ... and from round 6 the long-K 3x3 layers whose 256 x 256 tiles fill whole rounds on conv_pp_kernel (profiles/r06a_pp_first_contact.txt)
    assert _plan(64, 256, 31, 256, 3, pad=2, dil=2)[:2] == ("pp", (256, 256))  # l3.c2: 241 tiles = 0.94 of one round
    assert _plan(64, 512, 31, 1024, 3, pad=1)[:2] == ("pp", (256, 256))        # l3.0.ds: 964 tiles = 0.94 of four rounds
    assert _plan(64, 256, 63, 512, 3, stride=2)[:2] == ("pp", (256, 256))      # l2.0.ds
    assert _plan(32, 256, 31, 256, 3, pad=2, dil=2)[0] != "pp"                 # B = 32: 121 tiles, a partial round of long tiles
    assert _plan(64, 256, 31, 1024, 1, res=True)[:2] == ("wreg", (128, 256))   # l3.c3
    assert _plan(64, 256, 31, 768, 3)[:2] == ("wreg", (128, 256))              # conv_search: 633 tiles of 256 x 256 = 2.47 rounds (0.82 of three): stays
    assert _plan(64, 512, 31, 128, 1)[0] == "igemm"                            # l2.c1
    assert _plan(64, 256, 63, 64, 1)[0] == "igemm"                             # l1.c1
    assert _plan(64, 128, 31, 128, 3, pad=1)[0] == "halo"                      # l2.c2
    # fp32: no register-fed kernel, no sequences
    assert _plan(8, 1024, 31, 256, 1, dtype="f32")[0] == "igemm" and _plan(8, 1024, 31, 256, 1, dtype="f32")[2] == -1
