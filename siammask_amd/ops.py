"""Per-op entry points of libsiammask_hip.so on torch CUDA(HIP) tensors (unit parity tests).

conv2d    -> the implicit-GEMM MFMA kernel every convolution of the path uses
dw_xcorr  -> models/rpn.py:32-38 conv2d_dw_group
maxpool   -> nn.MaxPool2d(3, 2, 1) (experiments/siammask_sharp/resnet.py:158)
All take / return float32 NCHW tensors; ``dtype`` selects the device arithmetic type."""
import ctypes

import numpy as np
import torch

from . import _lib

ALGO = {"mfma": 0, "naive": 1, "mfma_nchw": 2, "naive_nchw": 3, "halo": 4, "wreg": 5, "pp": 6}
WREG_TILE = {(64, 256): 1, (64, 128): 2, (64, 64): 3, (128, 256): 4, (128, 128): 5, (128, 64): 6, (96, 256): 7, (32, 64): 8}
TILE = {None: 0, "auto": 0, (128, 128): 1, (128, 64): 2, (64, 128): 3, (64, 64): 4, (256, 128): 5}


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("siammask_amd.ops run on the MI355X only (got a CPU tensor)")


def tile_code(tile=None, kt=0, stages=0):
    """second byte of `algo`: tile override, K-tile bytes (0|128|256), LDS ring depth (0|2|3|4)"""
    return TILE[tile] | ({0: 0, 128: 1, 256: 2}[kt] << 4) | ({0: 0, 2: 1, 3: 2, 4: 3}[stages] << 6)


def wreg_code(tile, stages=0):
    """second byte of `algo` for algo='wreg': conv_wreg_kernel tile (bm, bn) and A-ring depth (0|3|4)"""
    return WREG_TILE[tuple(tile)] | ({0: 0, 8: 1, 3: 2, 4: 3}[stages] << 6)


def conv2d(x, w, b=None, stride=1, pad=0, dil=1, relu=False, res=None, res_mode=1, dtype="f32",
           algo="mfma", tile=None, win=None, ups=None, pos=None, pos_mul=0, pos_add=0, org=(0, 0),
           cin_off=0, cin_len=0, kt=0, stages=0):
    _chk_cuda(x, res)
    x = x.contiguous().float()
    B, Cin, H, W = x.shape
    w = np.ascontiguousarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else w, dtype=np.float32)
    bb = None if b is None else np.ascontiguousarray(
        b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, dtype=np.float32)
    g = _lib.ConvGeom()
    g.B, g.Cin, g.H, g.W = B, Cin, H, W
    g.Cout, g.k, g.stride, g.pad, g.dil = w.shape[0], w.shape[2], stride, pad, dil
    g.relu = int(bool(relu))
    g.res_mode = res_mode if res is not None else 0
    g.cin_off, g.cin_len = cin_off, cin_len
    g.org_y, g.org_x = org
    g.pos_mul, g.pos_add = pos_mul, pos_add
    Hl, Wl = H, W
    if win is not None:
        g.win, g.Hl, g.Wl = 1, win[0], win[1]
        Hl, Wl = win
    if ups is not None:
        g.ups, g.Hl, g.Wl = 1, ups[0], ups[1]
        Hl, Wl = ups
    Ho = (Hl + 2 * pad - dil * (g.k - 1) - 1) // stride + 1
    Wo = (Wl + 2 * pad - dil * (g.k - 1) - 1) // stride + 1
    y = torch.empty((B, g.Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    p = None if pos is None else np.ascontiguousarray(pos, dtype=np.int32)
    r = None if res is None else res.contiguous().float()
    code = ALGO[algo] | ((wreg_code(tile, stages) if algo == "wreg" else tile_code(tile, kt, stages)) << 8)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().smk_op_conv2d_ex(
            _lib.DTYPE[dtype], code, ctypes.byref(g), x.data_ptr(), vp(w), vp(bb),
            r.data_ptr() if r is not None else None, vp(p), y.data_ptr(), _lib.current_stream_ptr()))
    return y


SEQ_CFG = {None: -1, (64, 256): 0, (64, 128): 1, (64, 64): 2, (128, 256): 3, (128, 128): 4, "deep": 5,
           "no_a": 6, "no_w": 7, "no_mfma": 8, (128, 64): 9,
           "abl10": 10, "abl11": 11, "abl12": 12, "abl13": 13, "abl14": 14, "old64x128": 15, "old128x256": 16, "old64x256": 17, "old64x64": 18,      # 6..8: measurement builds of the 64x128 tile (wrong results by construction)
           "halo128": 24, "halo64": 25}     # 3x3 stride-1: whole-row tiles (128 / 64 pixels x 64 channels), activation patch shared by the nine taps


def conv_seq(x, layers, iters=1, want_outputs=True, info=None):
    """smk_op_conv_seq: a list of convolutions as ONE persistent conv_seq_kernel launch (fp16).

    x: [B,C,H,W] float32 CUDA tensor.  layers: dicts with w [Cout,Cin,k,k] (numpy), optional b, stride, pad, dil, relu,
    src (-1 = x, j = output of layer j; default: the previous layer), res (source index of the residual, -1 = x) with
    res_mode 1 (before the ReLU) / 2 (after), sync (default True), tile ((bm, bn), "deep" or None), kstag (-1 engine's choice).
    Returns (outputs [list of float32 NCHW tensors], usec per launch, per-layer (tiles_us, arrive_us) array); info (a dict,
    optional) receives "fused_pairs" = the (conv3, next 1x1) pairs the launch ran as one tile routine (smk_tune "seq_fuse").
    want_outputs: True (all layers), False, or a collection of layer indices (the returned list then holds None elsewhere)."""
    _chk_cuda(x)
    x = x.contiguous().float()
    B = x.shape[0]
    n = len(layers)
    arr = (_lib.SeqOp * n)()
    keep = []
    shapes = [tuple(x.shape[1:])]          # shape of x, then of every output
    outs = []
    for i, l in enumerate(layers):
        w = np.ascontiguousarray(l["w"], dtype=np.float32)
        b = None if l.get("b") is None else np.ascontiguousarray(l["b"], dtype=np.float32)
        keep += [w, b]
        src = l.get("src", i - 1)
        cin, H, W = shapes[src + 1]
        k, stride, pad, dil = w.shape[2], l.get("stride", 1), l.get("pad", 0), l.get("dil", 1)
        g = arr[i].g
        g.B, g.Cin, g.H, g.W = B, cin, H, W
        g.Cout, g.k, g.stride, g.pad, g.dil = w.shape[0], k, stride, pad, dil
        g.relu = int(bool(l.get("relu", False)))
        g.res_mode = l.get("res_mode", 1) if "res" in l else 0
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        shapes.append((w.shape[0], Ho, Wo))
        arr[i].src = src
        arr[i].res_src = l.get("res", -2)
        arr[i].sync = int(bool(l.get("sync", True)))
        arr[i].cfg = SEQ_CFG[l.get("tile")]
        arr[i].kstag = l.get("kstag", -1)
        arr[i].w_host = w.ctypes.data
        arr[i].b_host = b.ctypes.data if b is not None else None
        if want_outputs is True or (want_outputs not in (False, None) and i in want_outputs):
            y = torch.empty((B, w.shape[0], Ho, Wo), dtype=torch.float32, device=x.device)
            outs.append(y)
            arr[i].y_dev = y.data_ptr()
        else:
            if want_outputs not in (True, False, None):
                outs.append(None)              # (a collection of layer indices: the other layers' outputs are not read back)
            arr[i].y_dev = None
    us = ctypes.c_float(0.0)
    clk = np.zeros(2 * n, dtype=np.float32)
    nfused = ctypes.c_int(0)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().smk_op_conv_seq(arr, n, x.data_ptr(), iters, ctypes.byref(us),
                                              clk.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nfused), _lib.current_stream_ptr()))
    if info is not None:
        info["fused_pairs"] = nfused.value
    return outs, us.value, clk.reshape(n, 2)


def dw_xcorr(x, k, dtype="f32"):
    _chk_cuda(x, k)
    x, k = x.contiguous().float(), k.contiguous().float()
    B, C, H, W = x.shape
    kh, kw = k.shape[2], k.shape[3]
    y = torch.empty((B, C, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().smk_op_dw_xcorr(_lib.DTYPE[dtype], x.data_ptr(), k.data_ptr(), B, C, H, W, kh, kw,
                                              y.data_ptr(), _lib.current_stream_ptr()))
    return y


def maxpool3x3s2(x, dtype="f32"):
    _chk_cuda(x)
    x = x.contiguous().float()
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().smk_op_maxpool3x3s2(_lib.DTYPE[dtype], x.data_ptr(), B, C, H, W, y.data_ptr(),
                                                  _lib.current_stream_ptr()))
    return y


def bench_conv(B, Cin, H, W, Cout, k, stride=1, pad=0, dil=1, dtype="f16", tile=None, kt=0, stages=0,
               res=False, nchw=False, win=None, pos_mul=0, pos_add=0, iters=50, halo=False, wreg=False, pp=False):
    """average microseconds per launch of the MFMA conv kernel on this geometry (smk_bench_conv)"""
    g = _lib.ConvGeom()
    g.B, g.Cin, g.H, g.W = B, Cin, H, W
    g.Cout, g.k, g.stride, g.pad, g.dil = Cout, k, stride, pad, dil
    g.relu = 1
    g.pos_mul, g.pos_add = pos_mul, pos_add
    if win is not None:
        g.win, g.Hl, g.Wl = 1, win[0], win[1]
    us = ctypes.c_float(0.0)
    if pp:
        code = 6
    elif wreg:
        code = 5 | (wreg_code(tile, stages) << 8)
    else:
        code = (4 if halo else (2 if nchw else 0)) | (tile_code(tile, kt, stages) << 8)
    _lib.check(_lib.lib().smk_bench_conv(_lib.DTYPE[dtype], code, ctypes.byref(g), int(bool(res)), iters,
                                         ctypes.byref(us), _lib.current_stream_ptr()))
    return us.value
