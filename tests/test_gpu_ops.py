"""GPU parity of the individual HIP kernels against the CPU oracle (oracle/np_oracle.py),
called through the C ABI (smk_op_*).  Tolerances (rel. to max|ref|):
  fp32 path : 2e-5   (exact-fp32 MFMA fma chains, K <= 4608; oracle in float64)
  fp16 path : 2e-3   (inputs/weights rounded to fp16 first, fp32 accumulate, fp16 store)
Every distinct convolution class of SURVEY.md Appendix D is covered at reduced channel
counts plus the real heavy shapes."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = {"f32": 2e-5, "f16": 2e-3}


def _ops():
    from siammask_amd import ops
    return ops


def _q(a, dtype):
    """round to the device storage type (the kernels see fp16-rounded inputs in f16 mode)"""
    return a.astype(np.float16).astype(np.float64) if dtype == "f16" else a.astype(np.float64)


def _rand(rng, *shape):
    return rng.uniform(-1, 1, size=shape).astype(np.float32)


CONV_CLASSES = [
    # cin, cout, k, stride, pad, dil, hw, B
    (3, 64, 7, 2, 0, 1, 63, 2),       # stem (Cin=3 -> 8)
    (64, 64, 1, 1, 0, 1, 31, 2),      # l1 1x1
    (64, 64, 3, 1, 1, 1, 31, 2),      # l1 3x3 p1
    (128, 128, 3, 2, 0, 1, 31, 1),    # l2.0 3x3 s2 p0
    (256, 256, 3, 1, 2, 2, 15, 1),    # l3 3x3 d2 p2
    (256, 256, 3, 1, 0, 1, 15, 1),    # conv_search 3x3 p0
    (512, 96, 3, 1, 1, 1, 9, 3),      # big K, odd N
    (256, 10, 1, 1, 0, 1, 25, 2),     # cls head (N=10)
    (32, 16, 3, 1, 1, 1, 15, 2),      # refine small
    (4, 1, 3, 1, 1, 1, 21, 2),        # refine tail Cin=4 Cout=1
]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cfg", CONV_CLASSES)
def test_conv_classes(cfg, dtype):
    ops = _ops()
    cin, cout, k, stride, pad, dil, hw, B = cfg
    rng = np.random.default_rng(hash(cfg) & 0xffff)
    x, w, b = _rand(rng, B, cin, hw, hw), _rand(rng, cout, cin, k, k) / np.sqrt(cin * k * k), _rand(rng, cout)
    ref = O.conv2d(_q(x, dtype), _q(w, dtype), b.astype(np.float64), stride, pad, dil)
    xd = torch.from_numpy(x).cuda()
    errs = {}
    for algo in ("naive", "mfma", "mfma_nchw", "naive_nchw"):
        y = ops.conv2d(xd, w, b, stride, pad, dil, dtype=dtype, algo=algo).cpu().numpy()
        errs[algo] = rel_err(y, ref)
    # every instantiation of the kernel: workgroup tile x K-tile bytes x LDS ring depth x epilogue
    for tile, kt in (((128, 128), 128), ((128, 128), 256), ((128, 64), 128), ((128, 64), 256),
                     ((64, 128), 128), ((64, 128), 256), ((64, 64), 256), ((256, 128), 128)):
        for stages in (2, 3, 4):
            for algo in ("mfma", "mfma_nchw"):
                y = ops.conv2d(xd, w, b, stride, pad, dil, dtype=dtype, algo=algo, tile=tile, kt=kt,
                               stages=stages).cpu().numpy()
                errs["%s%s/%d/s%d" % (algo, tile, kt, stages)] = rel_err(y, ref)
    bad = {a: e for a, e in errs.items() if not e <= TOL[dtype]}
    assert not bad, "conv %s %s: %s (all: %s)" % (cfg, dtype, bad, errs)


HALO_CASES = [
    # cin, cout, pad, dil, hw, B   (3x3 stride 1; Cin a multiple of the 128-byte channel chunk)
    (64, 64, 1, 1, 31, 2),        # l1/l2-style 3x3 p1
    (128, 96, 1, 1, 33, 1),       # odd N, odd size
    (256, 256, 2, 2, 31, 1),      # l3 dilated
    (256, 128, 0, 1, 31, 2),      # conv_search: p0 (29x29 out)
    (64, 32, 1, 1, 63, 1),        # wide image: many row wraps per tile
    (512, 256, 1, 1, 15, 3),      # small image, long K, tiles end inside the image
]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cfg", HALO_CASES)
def test_conv3x3_halo_kernel(cfg, dtype):
    """conv3x3_halo_kernel (activation patch shared by the nine taps, chunk-major weight pack) against the
    oracle, both workgroup shapes, with bias + ReLU + residual."""
    ops = _ops()
    cin, cout, pad, dil, hw, B = cfg
    rng = np.random.default_rng(hash(cfg) & 0xffff)
    x, w, b = _rand(rng, B, cin, hw, hw), _rand(rng, cout, cin, 3, 3) / np.sqrt(cin * 9), _rand(rng, cout)
    ho = hw + 2 * pad - 2 * dil
    res = _rand(rng, B, cout, ho, ho)
    ref = np.maximum(O.conv2d(_q(x, dtype), _q(w, dtype), b.astype(np.float64), 1, pad, dil) + _q(res, dtype), 0)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(res).cuda()
    for tile in ((128, 128), (64, 128)):
        y = ops.conv2d(xd, w, b, 1, pad, dil, relu=True, res=rd, res_mode=1, dtype=dtype, algo="halo", tile=tile)
        e = rel_err(y.cpu().numpy(), ref)
        assert e <= TOL[dtype], "halo %s %s tile %s: %.3e" % (cfg, dtype, tile, e)


def test_conv3x3_halo_windows():
    """per-stream windows (Refine v*.0: F.pad + slice at a position) through the halo kernel"""
    ops = _ops()
    rng = np.random.default_rng(6)
    f = _rand(rng, 3, 64, 31, 31)
    w = _rand(rng, 32, 64, 3, 3) / 24
    pos = np.array([[0, 24], [12, 12], [24, 3]], dtype=np.int32)
    for dtype in ("f32", "f16"):
        fp = np.pad(_q(f, dtype), ((0, 0), (0, 0), (4, 4), (4, 4)))
        ref = np.concatenate([O.conv2d(fp[b:b + 1, :, y:y + 15, x:x + 15], _q(w, dtype), None, 1, 1, 1)
                              for b, (y, x) in enumerate(pos)])
        y = ops.conv2d(torch.from_numpy(f).cuda(), w, pad=1, win=(15, 15), pos=pos, pos_mul=1, pos_add=-4, dtype=dtype,
                       algo="halo", tile=(64, 128))
        assert rel_err(y.cpu().numpy(), ref) <= TOL[dtype]


SPLITK_CASES = [
    # cin, cout, k, pad, hw, B     K tiles (f16, 128-byte tile): 36 / 4 / 72 / 16
    (256, 96, 3, 1, 15, 2),
    (256, 64, 1, 0, 13, 1),
    (512, 128, 3, 1, 9, 1),
    (1024, 256, 1, 0, 15, 1),
]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cfg", SPLITK_CASES)
def test_conv_split_k(cfg, dtype):
    """split-K across workgroups (partial tiles in scratch, last-arrival reduction, counters self-resetting):
    forced 2- and 4-way splits on every tile shape that allows them, against the oracle; bias + ReLU + residual.
    The second call of each configuration re-uses the counters the first one left behind."""
    from siammask_amd import _lib
    ops = _ops()
    cin, cout, k, pad, hw, B = cfg
    rng = np.random.default_rng(hash(cfg) & 0xffff)
    x, w, b = _rand(rng, B, cin, hw, hw), _rand(rng, cout, cin, k, k) / np.sqrt(cin * k * k), _rand(rng, cout)
    ho = hw + 2 * pad - (k - 1)
    res = _rand(rng, B, cout, ho, ho)
    ref = np.maximum(O.conv2d(_q(x, dtype), _q(w, dtype), b.astype(np.float64), 1, pad, 1) + _q(res, dtype), 0)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(res).cuda()
    try:
        for sp in (2, 4):
            _lib.tune(ksplit=sp)
            for tile in ((64, 64), (64, 128), (128, 64), (128, 128)):
                for rep in range(2):
                    y = ops.conv2d(xd, w, b, 1, pad, 1, relu=True, res=rd, res_mode=1, dtype=dtype, tile=tile)
                    e = rel_err(y.cpu().numpy(), ref)
                    assert e <= TOL[dtype], "split-K x%d %s %s tile %s rep %d: %.3e" % (sp, cfg, dtype, tile, rep, e)
    finally:
        _lib.tune(ksplit=0)              # the library default


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_conv_epilogues(dtype):
    """bias + residual (before / after ReLU) + ReLU, as used by Bottleneck (resnet.py:99-101)
    and Refine's h(out)+v(p) sums (custom.py:150-152)."""
    ops = _ops()
    rng = np.random.default_rng(3)
    x, w, b = _rand(rng, 2, 64, 17, 17), _rand(rng, 96, 64, 1, 1) / 8, _rand(rng, 96)
    res = _rand(rng, 2, 96, 17, 17)
    base = O.conv2d(_q(x, dtype), _q(w, dtype), b.astype(np.float64))
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(res).cuda()
    for algo, tile in (("naive", None), ("mfma", None), ("mfma", (128, 128)), ("mfma", (128, 64)),
                       ("mfma", (64, 128)), ("mfma", (64, 64)), ("mfma", (256, 128))):
        y = ops.conv2d(xd, w, b, relu=True, dtype=dtype, algo=algo, tile=tile).cpu().numpy()
        assert rel_err(y, np.maximum(base, 0)) <= TOL[dtype], (algo, tile)
        y = ops.conv2d(xd, w, b, relu=True, res=rd, res_mode=1, dtype=dtype, algo=algo, tile=tile).cpu().numpy()
        assert rel_err(y, np.maximum(base + _q(res, dtype), 0)) <= TOL[dtype], (algo, tile)
        y = ops.conv2d(xd, w, b, relu=True, res=rd, res_mode=2, dtype=dtype, algo=algo, tile=tile).cpu().numpy()
        assert rel_err(y, np.maximum(base, 0) + _q(res, dtype)) <= TOL[dtype], (algo, tile)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_conv_window_upsample_slice(dtype):
    ops = _ops()
    rng = np.random.default_rng(5)
    f = _rand(rng, 3, 64, 31, 31)
    w = _rand(rng, 32, 64, 3, 3) / 24
    pos = np.array([[0, 24], [12, 12], [24, 3]], dtype=np.int32)
    fp = np.pad(_q(f, dtype), ((0, 0), (0, 0), (4, 4), (4, 4)))
    ref = np.concatenate([O.conv2d(fp[b:b + 1, :, y:y + 15, x:x + 15], _q(w, dtype), None, 1, 1, 1)
                          for b, (y, x) in enumerate(pos)])
    fd = torch.from_numpy(f).cuda()
    for algo, tile in (("naive", None), ("mfma", None), ("mfma", (128, 64)), ("mfma", (64, 64))):
        y = ops.conv2d(fd, w, pad=1, win=(15, 15), pos=pos, pos_mul=1, pos_add=-4, dtype=dtype, algo=algo, tile=tile)
        assert rel_err(y.cpu().numpy(), ref) <= TOL[dtype], (algo, tile)
    x = _rand(rng, 2, 32, 15, 15)
    w2 = _rand(rng, 16, 32, 3, 3) / 17
    ref = O.conv2d(O.upsample_nearest(_q(x, dtype), 31), _q(w2, dtype), None, 1, 1, 1)
    for algo in ("naive", "mfma", "mfma_nchw"):
        y = ops.conv2d(torch.from_numpy(x).cuda(), w2, pad=1, ups=(31, 31), dtype=dtype, algo=algo)
        assert rel_err(y.cpu().numpy(), ref) <= TOL[dtype], algo
    x = _rand(rng, 2, 96, 9, 9)
    w3 = _rand(rng, 24, 32, 1, 1) / 6
    ref = O.conv2d(_q(x[:, 32:64], dtype), _q(w3, dtype))
    y = ops.conv2d(torch.from_numpy(x).cuda(), w3, cin_off=32, cin_len=32, dtype=dtype)
    assert rel_err(y.cpu().numpy(), ref) <= TOL[dtype]
    # template centre crop folded into the adjust conv (custom.py:21-24)
    x = _rand(rng, 2, 64, 15, 15)
    w4 = _rand(rng, 32, 64, 1, 1) / 8
    ref = O.conv2d(_q(x, dtype), _q(w4, dtype))[:, :, 4:-4, 4:-4]
    y = ops.conv2d(torch.from_numpy(x).cuda(), w4, win=(7, 7), org=(4, 4), dtype=dtype)
    assert rel_err(y.cpu().numpy(), ref) <= TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_heavy_real_shapes(dtype):
    """the two heaviest GEMM classes at their real size, B=1."""
    ops = _ops()
    rng = np.random.default_rng(11)
    for cin, cout, k, pad, dil in ((512, 1024, 3, 1, 1), (1024, 256, 1, 0, 1)):
        x = _rand(rng, 1, cin, 31, 31)
        w = _rand(rng, cout, cin, k, k) / np.sqrt(cin * k * k)
        ref = O.conv2d(_q(x, dtype), _q(w, dtype), None, 1, pad, dil)
        xd = torch.from_numpy(x).cuda()
        for tile, kt, stages in ((None, 0, 0), ((128, 128), 128, 2), ((128, 128), 128, 4), ((128, 64), 256, 3),
                                 ((64, 128), 128, 4), ((64, 64), 256, 2), ((64, 64), 256, 4), ((256, 128), 128, 3)):
            y = ops.conv2d(xd, w, pad=pad, dil=dil, dtype=dtype, tile=tile, kt=kt, stages=stages).cpu().numpy()
            assert rel_err(y, ref) <= TOL[dtype], (cin, tile, kt, stages)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("shape", [(2, 256, 29, 29, 5, 5), (1, 768, 29, 29, 5, 5), (3, 64, 12, 9, 3, 2)])
def test_dw_xcorr(shape, dtype):
    """models/rpn.py:32-38 conv2d_dw_group; includes a ragged (non-square, small tap) case."""
    ops = _ops()
    B, C, H, W, kh, kw = shape
    rng = np.random.default_rng(17)
    x, k = _rand(rng, B, C, H, W), _rand(rng, B, C, kh, kw)
    ref = O.conv2d_dw_group(_q(x, dtype), _q(k, dtype))
    y = ops.dw_xcorr(torch.from_numpy(x).cuda(), torch.from_numpy(k).cuda(), dtype=dtype).cpu().numpy()
    assert y.shape == ref.shape
    assert rel_err(y, ref) <= TOL[dtype]


def test_dw_xcorr_linearity_full_batch():
    """size-independent property at the BASELINE batch (64): xcorr is bilinear."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(0)
    x1, x2 = torch.randn(64, 768, 29, 29, generator=g).cuda(), torch.randn(64, 768, 29, 29, generator=g).cuda()
    k = torch.randn(64, 768, 5, 5, generator=g).cuda()
    a = ops.dw_xcorr(x1, k) + 2.0 * ops.dw_xcorr(x2, k)
    b = ops.dw_xcorr(x1 + 2.0 * x2, k)
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_maxpool(dtype):
    ops = _ops()
    rng = np.random.default_rng(19)
    for hw in (125, 61, 8):
        x = np.maximum(_rand(rng, 2, 64, hw, hw), 0)   # post-ReLU input as on the path
        ref = O.maxpool_3x3_s2_p1(_q(x, dtype))
        y = ops.maxpool3x3s2(torch.from_numpy(x).cuda(), dtype=dtype).cpu().numpy()
        assert y.shape == ref.shape and rel_err(y, ref) == 0.0   # max is exact


WREG_TILES = ((64, 256), (64, 128), (64, 64), (128, 256), (128, 128), (128, 64), (96, 256), (32, 64))
WREG_CASES = [
    # cin, cout, k, stride, pad, dil, hw, B, residual
    (64, 64, 1, 1, 0, 1, 31, 2, False),      # short K (one 128-element pad: two K tiles)
    (256, 1024, 1, 1, 0, 1, 15, 1, True),    # bottleneck conv3 + residual + ReLU, wide N
    (1024, 256, 1, 1, 0, 1, 15, 1, False),   # bottleneck conv1, long K
    (256, 256, 3, 1, 2, 2, 15, 1, False),    # l3 3x3 d2 p2 (tap-uniform K tiles)
    (128, 128, 3, 2, 0, 1, 31, 1, False),    # 3x3 s2 p0
    (512, 96, 3, 1, 1, 1, 9, 3, True),       # long K, N = 96: tile overhang beyond the packed rows (SRD zero fill)
    (256, 10, 1, 1, 0, 1, 25, 2, False),     # N = 10
    (32, 16, 3, 1, 1, 1, 15, 2, False),      # Ci < K tile: taps straddle K tiles (per-lane tap decode)
    (3, 64, 7, 2, 0, 1, 63, 2, False),       # Cin 3 -> 8, K = 392 -> 512
]


@pytest.mark.parametrize("cfg", WREG_CASES)
def test_conv_wreg_kernel(cfg):
    """conv_wreg_kernel (weights in MFMA-fragment order straight into registers, activations through LDS) against the
    oracle: every workgroup shape x the three prefetch depths (bit-equal among themselves), bias + ReLU (+ residual), M tails, N overhang."""
    ops = _ops()
    cin, cout, k, stride, pad, dil, hw, B, with_res = cfg
    rng = np.random.default_rng(hash(cfg) & 0xffff)
    x, w, b = _rand(rng, B, cin, hw, hw), _rand(rng, cout, cin, k, k) / np.sqrt(cin * k * k), _rand(rng, cout)
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ref = O.conv2d(_q(x, "f16"), _q(w, "f16"), b.astype(np.float64), stride, pad, dil)
    rd = None
    if with_res:
        res = _rand(rng, B, cout, ho, ho)
        ref = ref + _q(res, "f16")
        rd = torch.from_numpy(res).cuda()
    ref = np.maximum(ref, 0)
    xd = torch.from_numpy(x).cuda()
    errs = {}
    for tile in WREG_TILES:
        ys = {}
        for stages in (3, 4, 8):             # (8 = the measured-slower deep prefetch: its own kernels in `make MEASURE=1` builds, the 3-deep ring otherwise)
            y = ops.conv2d(xd, w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16", algo="wreg", tile=tile,
                           stages=stages)
            ys[stages] = y.clone()
            errs["%dx%d/s%d" % (tile[0], tile[1], stages)] = rel_err(y.cpu().numpy(), ref)
        # how far ahead the operand streams run (ring depth, weight fragments in registers) changes no accumulator's k order
        assert torch.equal(ys[3], ys[4]) and torch.equal(ys[3], ys[8]), (cfg, tile)
    bad = {a: e for a, e in errs.items() if not e <= TOL["f16"]}
    assert not bad, "wreg %s: %s (all: %s)" % (cfg, bad, errs)


def test_conv_wreg_windows_and_upsample():
    """the gather features the Refine convolutions use (per-stream windows, nearest upsampling) through conv_wreg_kernel"""
    ops = _ops()
    rng = np.random.default_rng(16)
    f = _rand(rng, 3, 64, 31, 31)
    w = _rand(rng, 32, 64, 3, 3) / 24
    pos = np.array([[0, 24], [12, 12], [24, 3]], dtype=np.int32)
    fp = np.pad(_q(f, "f16"), ((0, 0), (0, 0), (4, 4), (4, 4)))
    ref = np.concatenate([O.conv2d(fp[b:b + 1, :, y:y + 15, x:x + 15], _q(w, "f16"), None, 1, 1, 1)
                          for b, (y, x) in enumerate(pos)])
    for tile in WREG_TILES:
        y = ops.conv2d(torch.from_numpy(f).cuda(), w, pad=1, win=(15, 15), pos=pos, pos_mul=1, pos_add=-4, dtype="f16",
                       algo="wreg", tile=tile)
        assert rel_err(y.cpu().numpy(), ref) <= TOL["f16"], tile
    g = _rand(rng, 2, 32, 15, 15)
    w2 = _rand(rng, 16, 32, 3, 3) / 17
    ref2 = O.conv2d(O.upsample_nearest(_q(g, "f16"), 31), _q(w2, "f16"), None, 1, 1, 1)
    for tile in ((64, 64), (128, 64)):
        y = ops.conv2d(torch.from_numpy(g).cuda(), w2, pad=1, ups=(31, 31), dtype="f16", algo="wreg", tile=tile)
        assert rel_err(y.cpu().numpy(), ref2) <= TOL["f16"], tile


PRODUCER_VARIANTS = ((0, 2), (1, 2), (0, 4), (1, 4))       # (a_stage, npw); the first one is the reference


def test_conv_wreg_producer_variants_bit_equal():
    """smk_tune a_stage / npw change only HOW the activation rows reach the LDS ring of conv_wreg_kernel: by LDS-DMA with
    the swizzle on the source address or through registers in ascending lane order with the swizzle applied by
    ds_write_b128 (a_stage), issued by two or by four producer waves (npw).  The LDS image and everything behind it are
    the same, so the outputs must be bit-identical for every workgroup shape and geometry (padding, strides, dilation, taps
    straddling K tiles, per-stream windows, M tails, N overhang)."""
    from siammask_amd import _lib
    ops = _ops()
    rng = np.random.default_rng(77)
    saved = {k: _lib.tune_get(k) for k in ("a_stage", "npw")}
    try:
        for cfg in WREG_CASES:
            cin, cout, k, stride, pad, dil, hw, B, with_res = cfg
            x, w, b = _rand(rng, B, cin, hw, hw), _rand(rng, cout, cin, k, k) / np.sqrt(cin * k * k), _rand(rng, cout)
            ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
            rd = torch.from_numpy(_rand(rng, B, cout, ho, ho)).cuda() if with_res else None
            xd = torch.from_numpy(x).cuda()
            for tile in WREG_TILES:
                ys = []
                for a, n in PRODUCER_VARIANTS:
                    _lib.tune(a_stage=a, npw=n)
                    ys.append(ops.conv2d(xd, w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16",
                                         algo="wreg", tile=tile, stages=3).clone())
                for v, y in zip(PRODUCER_VARIANTS[1:], ys[1:]):
                    assert torch.equal(ys[0], y), (cfg, tile, v)
        f = _rand(rng, 3, 64, 31, 31)
        w = _rand(rng, 32, 64, 3, 3) / 24
        pos = np.array([[0, 24], [12, 12], [24, 3]], dtype=np.int32)
        for tile in WREG_TILES:
            ys = []
            for a, n in PRODUCER_VARIANTS:
                _lib.tune(a_stage=a, npw=n)
                ys.append(ops.conv2d(torch.from_numpy(f).cuda(), w, pad=1, win=(15, 15), pos=pos, pos_mul=1, pos_add=-4,
                                     dtype="f16", algo="wreg", tile=tile).clone())
            for v, y in zip(PRODUCER_VARIANTS[1:], ys[1:]):
                assert torch.equal(ys[0], y), (tile, v)
    finally:
        _lib.tune(**saved)


PP_CASES = [
    # cin, cout, k, stride, pad, dil, hw, B, with_res      (M spans several 256-row tiles, the last one ragged)
    (256, 256, 3, 1, 2, 2, 31, 2, False),    # layer3 conv2: dilation 2 (resnet.py:69-72,165), 8 tiles, tail of 130 rows
    (256, 512, 3, 2, 0, 1, 63, 1, False),    # layer2.0 shortcut: 3x3 stride 2 pad 0 (resnet.py:195-206), two channel tiles
    (256, 768, 3, 1, 0, 1, 31, 1, False),    # conv_search x3 as one N = 768 GEMM (models/rpn.py:50-54), 29 x 29 outputs
    (512, 1024, 3, 1, 1, 1, 15, 2, False),   # layer3.0 shortcut: K = 4608 (72 K tiles), four channel tiles
    (64, 248, 3, 1, 1, 1, 25, 1, True),      # K = 576 -> 640 (a K tile of pure padding), N overhang, residual + ReLU
    (1024, 256, 1, 1, 0, 1, 20, 1, True),    # 1x1, long K, residual
    (128, 256, 1, 1, 0, 1, 9, 3, False),     # two K tiles only (the ring's prologue covers the whole K), M = 243 < one tile
]


@pytest.mark.parametrize("cfg", PP_CASES)
def test_conv_pp_kernel(cfg):
    """conv_pp_kernel (256 x 256 tiles, two wave groups alternating between LDS-DMA / fragment reads and MFMAs, both operands
    through LDS; the long-K convolutions of the B = 64 regime) against the oracle, and BIT-equal to conv_wreg_kernel's 128 x 256
    tile: same MFMA shape, same k order per accumulator, same epilogue arithmetic.  Run three times: a ring hazard would show
    as a rare wrong tile, not as a wrong kernel."""
    ops = _ops()
    cin, cout, k, stride, pad, dil, hw, B, with_res = cfg
    rng = np.random.default_rng(hash(cfg) & 0xffff)
    x, w, b = _rand(rng, B, cin, hw, hw), _rand(rng, cout, cin, k, k) / np.sqrt(cin * k * k), _rand(rng, cout)
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ref = O.conv2d(_q(x, "f16"), _q(w, "f16"), b.astype(np.float64), stride, pad, dil)
    rd = None
    if with_res:
        res = _rand(rng, B, cout, ho, ho)
        ref = ref + _q(res, "f16")
        rd = torch.from_numpy(res).cuda()
    ref = np.maximum(ref, 0)
    xd = torch.from_numpy(x).cuda()
    yw = ops.conv2d(xd, w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16", algo="wreg", tile=(128, 256), stages=3)
    for rep in range(3):
        y = ops.conv2d(xd, w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16", algo="pp")
        assert rel_err(y.cpu().numpy(), ref) <= TOL["f16"], (cfg, rep, rel_err(y.cpu().numpy(), ref))
        assert torch.equal(y, yw), (cfg, rep, float((y - yw).abs().max()))
