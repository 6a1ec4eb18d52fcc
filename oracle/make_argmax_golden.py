"""Test infrastructure (never imported by the product path): the fp64 oracle's best anchor index for the 1024 streams of the
argmax-agreement statistic (tools/measure/argmax_stats.py: B = 64 x 8 seeds x smooth / white-noise crops), computed ON THE CPU
and committed as tests/golden/argmax_oracle_1024.npz, so that the GPU test compares BOTH device dtypes with the reference's
answer on every stream without spending GPU-box minutes on 1024 float64 forward passes (about 25 min of an 8-core host).

The oracle (oracle/np_oracle.py) is pinned against the reference itself (tests/test_oracle_golden.py: <= 5e-7 of outputs
of /root/reference run in float64); the decode follows /root/reference/tools/test.py:205-254 (decode_best).

Per stream: the five best candidates (index, float64 pscore) -- enough to tell a genuine disagreement from a near-tie -- and
the target size the decode was given.

    python oracle/make_argmax_golden.py            # writes tests/golden/argmax_oracle_1024.npz
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle.np_oracle import Oracle, decode_best      # noqa: E402
from siammask_amd import synth                          # noqa: E402

KINDS = ("smooth", "noise")
B, SEEDS, CHUNK, TOP = 64, 8, 8, 5


def inputs(kind, seed, b0, n):
    """exactly the streams tools/measure/argmax_stats.collect feeds the device"""
    gen = {"smooth": synth.smooth_image_batch, "noise": synth.image_batch}[kind]
    s0 = 10000 * (seed + 1)
    z = gen(n, 127, stream0=s0 + b0)
    x = gen(n, 255, stream0=s0 + 5000 + b0)
    g = np.random.Generator(np.random.PCG64(7 + seed))
    twh = g.uniform(40.0, 110.0, size=(B, 2))[b0:b0 + n]
    return z, x, twh


def main(out=os.path.join(REPO, "tests", "golden", "argmax_oracle_1024.npz")):
    o = Oracle(synth.state_dict("sharp", "synthetic_damped"), "sharp")
    idx = np.zeros((len(KINDS), SEEDS, B, TOP), dtype=np.int16)
    val = np.zeros((len(KINDS), SEEDS, B, TOP), dtype=np.float64)
    twh_all = np.zeros((len(KINDS), SEEDS, B, 2), dtype=np.float64)
    t0 = time.time()
    for ki, kind in enumerate(KINDS):
        for seed in range(SEEDS):
            for b0 in range(0, B, CHUNK):
                z, x, twh = inputs(kind, seed, b0, CHUNK)
                o.template(z.astype(np.float64))
                cls, loc = o.track(x.astype(np.float64))[:2]
                for j in range(CHUNK):
                    ps = decode_best(cls[j], loc[j], target_sz=twh[j], scale_x=1.0)[3]
                    order = np.argsort(-ps, kind="stable")[:TOP]           # ties: lowest index first, like np.argmax
                    idx[ki, seed, b0 + j] = order
                    val[ki, seed, b0 + j] = ps[order]
                    twh_all[ki, seed, b0 + j] = twh[j]
            print("%s seed %d done (%.0f s)" % (kind, seed, time.time() - t0), flush=True)
    np.savez_compressed(out, kinds=np.array(KINDS), top_idx=idx, top_pscore=val, target_wh=twh_all,
                        note=np.array("fp64 oracle (oracle/np_oracle.py), fixture synthetic_damped, sharp; streams of "
                                      "tools/measure/argmax_stats.collect(B=64, seeds=8)"))
    print("wrote", out)


if __name__ == "__main__":
    main()
