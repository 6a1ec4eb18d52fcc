"""The build rules know every file a kernel object is made of (conv_seq.o did not depend on
wreg_halo_tile.inc, so an edit to the patch-sharing tile shipped a stale kernel)."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "siammask_amd", "csrc")


def local_includes(path, seen=None):
    """transitive `#include "x"` closure of a source file (paths relative to csrc/)"""
    seen = set() if seen is None else seen
    for m in re.finditer(r'^\s*#include\s+"([^"]+)"', open(path).read(), re.M):
        inc = os.path.normpath(os.path.join(os.path.dirname(path), m.group(1)))
        if inc not in seen and os.path.exists(inc):
            seen.add(inc)
            local_includes(inc, seen)
    return seen


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


@pytest.mark.parametrize("src", sources())
def test_every_include_triggers_a_rebuild(src):
    obj = os.path.splitext(src)[0] + ".o"
    incs = local_includes(os.path.join(CSRC, src))
    assert incs, "%s includes nothing local?" % src
    for inc in sorted(incs) + [os.path.join(CSRC, src)]:
        rel = os.path.relpath(inc, CSRC)
        # -W: pretend `rel` was just modified; -n: print what would run
        out = subprocess.run(["make", "-n", "-W", rel, obj], cwd=CSRC, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert src in out.stdout and "-c" in out.stdout, (
            "editing %s would NOT rebuild %s (Makefile prerequisites incomplete):\n%s" % (rel, obj, out.stdout))


def test_every_source_is_linked():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    objs = re.search(r"^OBJS\s*=\s*(.+)$", mk, re.M).group(1).split()
    assert sorted(objs) == sorted(os.path.splitext(s)[0] + ".o" for s in sources())
