"""siammask_amd -- MI355X-native per-frame inference path of SiamMask.

Host side (Python on PyTorch-ROCm) of libsiammask_hip.so: the drop-in ``Custom`` modules that
mirror the reference's ``experiments/*/custom.py`` call surface, the ctypes binding, the
synthetic-checkpoint generator and the multi-GPU stream sharding helper.
"""
__version__ = "0.1.0"
