"""Drop-in replacement for the reference's experiments/*/custom.py (sharp variant).

Put this directory on PYTHONPATH *ahead of* the reference experiment directory and the
unchanged tools (tools/test.py:559 `from custom import Custom`) pick up the MI355X path.
See INTEGRATION.md."""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

from siammask_amd.custom import CustomSharp as Custom  # noqa: E402,F401
