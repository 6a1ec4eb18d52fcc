#!/usr/bin/env python
"""sha256 over the kernel / engine sources of libsiammask_hip.so (what a PMC summary under profiles/ was measured on).
bench.py refuses a committed PMC summary whose hash differs from the sources it runs on."""
import glob
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_sources_sha256():
    h = hashlib.sha256()
    d = os.path.join(REPO, "siammask_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "*.cpp")) +
                    glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha256())
