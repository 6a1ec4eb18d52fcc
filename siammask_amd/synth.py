"""Deterministic synthetic checkpoints and inputs (there are no pretrained weights offline).

SURVEY.md section 8c: a raw random init is numerically useless (activations ~1e4, exp()
overflow in the host decode), so the synthetic checkpoint is *calibrated*: BN affine
parameters are perturbed so that folding is exercised, the residual branches are damped
(``bn3.weight *= damp`` as in trained ResNets) and the BN running statistics are set to the
batch statistics of one template + one search pass.  The calibrated running statistics were
computed once by ``oracle/make_golden.py`` *running the reference modules* and are
committed as ``tests/golden/bnstats_<fixture>.npz``; everything else is a pure function of
the numpy PCG64 stream, hence bit-identical on every machine.

Names/shapes come from ``siammask_amd.spec`` (the reference state-dict contract).
"""
import os
import zlib
from collections import OrderedDict

import numpy as np

from . import spec

_GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "tests", "golden")

FIXTURES = {
    # name: (seed, damp)  -- SURVEY.md 8c: damped is the default parity fixture,
    # stress (no damping, amplification ~200x) is for fp32 gates only.
    "synthetic_damped": (0, 0.25),
    "synthetic_stress": (0, 1.0),
}


def _rng_for(seed, name):
    # independent stream per tensor so that variants (rpn/base/sharp) share common tensors
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def raw_state_dict(variant="sharp", seed=0, damp=0.25):
    """Un-calibrated synthetic state dict: running_mean = 0, running_var = 1."""
    sd = OrderedDict()
    for name, (shape, kind) in spec.state_dict_spec(variant).items():
        g = _rng_for(seed, name)
        if kind == "conv_w":
            cout, cin, kh, kw = shape
            if name.startswith("refine_model.") or name.endswith("head.3.weight"):
                bound = np.sqrt(3.0 / (cin * kh * kw))          # variance-preserving uniform
                v = g.uniform(-bound, bound, size=shape)
            else:
                v = g.normal(0.0, np.sqrt(2.0 / (kh * kw * cout)), size=shape)
        elif kind == "deconv_w":
            cin = shape[0]
            bound = np.sqrt(3.0 / cin)
            v = g.uniform(-bound, bound, size=shape)
        elif kind == "bias":
            v = g.uniform(-0.1, 0.1, size=shape)
        elif kind == "bn_w":
            v = g.uniform(0.5, 1.0, size=shape)
            if name.endswith(".bn3.weight"):
                v = v * damp
        elif kind == "bn_b":
            v = g.normal(0.0, 0.1, size=shape)
        elif kind == "bn_mean":
            v = np.zeros(shape)
        elif kind == "bn_var":
            v = np.ones(shape)
        elif kind == "bn_nbt":
            sd[name] = np.array(1, dtype=np.int64)
            continue
        else:
            raise KeyError(kind)
        sd[name] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def bnstats_path(fixture):
    return os.path.join(_GOLDEN_DIR, "bnstats_%s.npz" % fixture)


def state_dict(variant="sharp", fixture="synthetic_damped", calibrated=True):
    """Calibrated synthetic checkpoint as {name: np.ndarray(float32)}."""
    seed, damp = FIXTURES[fixture]
    sd = raw_state_dict(variant, seed, damp)
    if calibrated:
        path = bnstats_path(fixture)
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s missing: run oracle/make_golden.py (needs /root/reference)" % path)
        stats = np.load(path)
        for name in sd:
            if name.endswith("running_mean") or name.endswith("running_var"):
                sd[name] = np.ascontiguousarray(stats[name], dtype=np.float32)
    return sd


def torch_state_dict(variant="sharp", fixture="synthetic_damped", calibrated=True):
    import torch
    return OrderedDict((k, torch.from_numpy(np.array(v)))
                       for k, v in state_dict(variant, fixture, calibrated).items())


def image_batch(batch, size, stream0=0, seed=1234):
    """Synthetic crops: integers U{0..255} as float32 NCHW (matches im_to_torch,
    tools/test.py:61-64: raw BGR 0-255, no normalisation).  One PCG64 stream per video stream."""
    out = np.empty((batch, 3, size, size), dtype=np.float32)
    for b in range(batch):
        g = np.random.Generator(np.random.PCG64([seed, stream0 + b, size]))
        out[b] = g.integers(0, 256, size=(3, size, size)).astype(np.float32)
    return out


def smooth_image_batch(batch, size, stream0=0, seed=1234):
    """Low-frequency synthetic crops (blobs) -- better conditioned than white noise."""
    out = np.empty((batch, 3, size, size), dtype=np.float32)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    for b in range(batch):
        g = np.random.Generator(np.random.PCG64([seed + 7, stream0 + b, size]))
        img = np.zeros((3, size, size))
        for _ in range(12):
            cx, cy = g.uniform(0, size, 2)
            s = g.uniform(size / 16.0, size / 3.0)
            amp = g.uniform(-1, 1, size=(3, 1, 1))
            img += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))[None]
        img = (img - img.min()) / (img.max() - img.min() + 1e-12) * 255.0
        img += g.normal(0, 4.0, size=img.shape)
        out[b] = np.clip(np.rint(img), 0, 255).astype(np.float32)
    return out
