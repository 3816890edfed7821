"""vptr_amd -- MI355X-native (gfx950) implementation of the VPTR video-prediction hot path.

Importing the package loads libvptr_hip.so (hand-written HIP kernels behind a C ABI, include/vptr_hip.h); there is
no CPU or stock-PyTorch fallback for the hot path.  `vptr_amd.model` mirrors the reference's `model` package API.
"""
from . import _lib, ops  # noqa: F401  (raises ImportError if the HIP library is missing)
from . import model  # noqa: F401

__version__ = "0.1.0"
