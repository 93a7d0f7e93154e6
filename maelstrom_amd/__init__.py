"""maelstrom_amd — MI355X-native ensemble engine for Maelstrom's message-routing + node-execution hot path.

Layout: csrc/ (HIP kernels + the C-ABI, built into libmaelsim.so), _abi.py (ctypes mirror of
include/maelsim.h), engine.py (host mirror of the reference's test-map / history / checker surface).
"""
from . import _abi  # noqa: F401
