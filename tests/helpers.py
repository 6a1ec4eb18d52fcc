"""Shared helpers for the parity tests: golden-fixture loading and tolerance checks."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("sharp_damped_b2", "sharp_stress_b1", "base_damped_b1", "rpn_damped_b1")


def load_golden(name):
    g = np.load(os.path.join(GOLD, "golden_%s.npz" % name), allow_pickle=False)
    return {k: g[k] for k in g.files}


def rel_err(got, ref):
    """max|got-ref| / max|ref| (the tolerance form of SURVEY.md 8c)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def sampled_err(gold, key, full):
    """Compare a full tensor (any float dtype, NCHW) with the strided sample stored in a fixture."""
    stride = int(gold[key + "__stride"])
    ref = gold[key + "__vals"].astype(np.float64)
    a = np.asarray(full, dtype=np.float64)
    assert tuple(a.shape) == tuple(gold[key + "__shape"]), (key, a.shape, gold[key + "__shape"])
    got = a.ravel()[::stride]
    return float(np.abs(got - ref).max() / (float(gold[key + "__maxabs"]) + 1e-30))


def assert_close(got, ref, tol, what):
    e = rel_err(got, ref)
    assert e <= tol, "%s: rel-to-max error %.3e > %.1e" % (what, e, tol)
    return e
