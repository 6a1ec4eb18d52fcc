"""The address arithmetic of the two tile routines added to conv_seq_kernel in round 3 (c3c1_tile.inc: fused conv3 + next 1x1;
wreg_halo_tile.inc: 3x3 layers on whole-row tiles with a shared activation patch), restated lane by lane in Python
(tools/measure/emu/) and checked against plain matrix products / a direct dilated convolution: LDS layouts (XOR swizzle, 144-byte
pitch), MFMA fragment ownership, fragment-order weight packs, ragged tiles, stores that must not touch rows beyond the tile.
This pins the DESIGN of the index math on the CPU; the kernels themselves are held by tests/test_gpu_seq.py."""
import os
import runpy

import pytest

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "measure", "emu")


@pytest.mark.parametrize("script", ["emu_c3c1_tile.py", "emu_wreg_halo_tile.py", "emu_c3c1p_tile.py", "emu_c3c1s_tile.py", "emu_c2front_tile.py", "emu_conv_pp.py"])
def test_tile_index_arithmetic(script, capsys):
    runpy.run_path(os.path.join(EMU, script), run_name="__main__")      # the scripts assert; their table goes to stdout
    out = capsys.readouterr().out
    assert out.count("err") >= 4, out
