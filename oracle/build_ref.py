#!/usr/bin/env python
"""Build oracle/_ref: the REFERENCE ITSELF in compiled form, so that it can travel to the GPU box.

The reference's inference path is Python (no native code to compile into a .so), so "built from the sources where they
lie under /root/reference, outputs only into oracle/_ref/" means here:

  * every module the unchanged tools import (tools/test.py, tools/demo.py, utils/, models/, experiments/*/custom.py and
    resnet.py) byte-compiled by CPython's own compiler (py_compile) into SOURCELESS .pyc files -- the same code objects
    the interpreter would build from the .py files, nothing edited, nothing re-typed;
  * utils/pyvotkit/region (the reference's Cython extension) built into a .so from its own .pyx;
  * the data the tools read: experiments/*/config*.json and the demo frames data/tennis/*.jpg.

No reference SOURCE enters the repository: oracle/_ref/ is git-ignored (it still travels with gpurun, like the built
libsiammask_hip.so).  Runs only where /root/reference exists; __graft_entry__.build() calls it.  Consumers (all test /
measurement infrastructure, never the product path): tests/compat/shim.py (the unchanged tools driving the HIP path on
the MI355X, tests/test_gpu_tools.py), bench.py's cpu_baseline leg (kind "reference")."""
import glob
import os
import py_compile
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.environ.get("SIAMMASK_REFERENCE_SRC", "/root/reference")
DST = os.path.join(REPO, "oracle", "_ref", "reference")
EXPERIMENTS = ("siammask_sharp", "siammask_base", "siamrpn_resnet")


def _compile_tree(rel, recursive=True):
    n = 0
    root = os.path.join(SRC, rel)
    if os.path.isfile(root):
        files = [root]
    else:
        files = glob.glob(os.path.join(root, "**", "*.py"), recursive=True) if recursive else glob.glob(os.path.join(root, "*.py"))
    for f in files:
        r = os.path.relpath(f, SRC)
        if os.path.basename(f) == "setup.py":
            continue
        out = os.path.join(DST, r + "c")               # legacy sourceless layout: foo.pyc beside where foo.py would be
        os.makedirs(os.path.dirname(out), exist_ok=True)
        py_compile.compile(f, cfile=out, dfile=os.path.join("reference", r), doraise=True)
        n += 1
    return n


def build(force=False):
    """-> path of the built tree, or None when the reference is not present on this machine"""
    if not os.path.isfile(os.path.join(SRC, "tools", "test.py")):
        return DST if os.path.isfile(os.path.join(DST, "tools", "test.pyc")) else None
    stamp = os.path.join(DST, ".built")
    if os.path.isfile(stamp) and not force:
        return DST
    shutil.rmtree(DST, ignore_errors=True)
    n = _compile_tree("tools/test.py") + _compile_tree("tools/demo.py") + _compile_tree("utils") + _compile_tree("models")
    for e in EXPERIMENTS:
        n += _compile_tree(os.path.join("experiments", e), recursive=False)
        for cfg in glob.glob(os.path.join(SRC, "experiments", e, "config*.json")):
            shutil.copy(cfg, os.path.join(DST, "experiments", e, os.path.basename(cfg)))
    os.makedirs(os.path.join(DST, "data", "tennis"), exist_ok=True)
    for jpg in sorted(glob.glob(os.path.join(SRC, "data", "tennis", "*.jpg"))):
        shutil.copy(jpg, os.path.join(DST, "data", "tennis", os.path.basename(jpg)))
    # the Cython extension: built in a scratch directory by the harness, the .so kept beside the package
    sys.path.insert(0, REPO)
    os.environ["SIAMMASK_REFERENCE"] = SRC
    from tests.compat import shim
    try:
        so = shim._build_region()
        shutil.copy(so, os.path.join(DST, "utils", "pyvotkit", os.path.basename(so)))
    except Exception as e:  # noqa: BLE001 -- the two names are only called by track_vot; the harness stubs them
        print("build_ref: utils.pyvotkit.region not built (%s)" % e)
    for root, dirs, files in os.walk(DST):
        for d in dirs:
            os.chmod(os.path.join(root, d), 0o755)
        for f in files:
            os.chmod(os.path.join(root, f), 0o644)
    open(stamp, "w").write("%d modules compiled from %s\n" % (n, SRC))
    return DST


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print("oracle/_ref:", p)
