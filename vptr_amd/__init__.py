"""vptr_amd -- MI355X-native (gfx950) implementation of the VPTR video-prediction hot path.

`vptr_amd.model` mirrors the reference's `model` package API; `vptr_amd.ops` are the autograd wrappers over the C-ABI
HIP kernels of `libvptr_hip.so` (include/vptr_hip.h).  There is no CPU or stock-PyTorch fallback for the hot path:
importing `vptr_amd.ops` / `vptr_amd.model` / `vptr_amd.train` loads the shared library and raises if it is missing or
stale.  Only `vptr_amd.build` (the hipcc driver that produces the library) is importable without it.
"""
import importlib

__version__ = "0.1.0"
_LAZY = ("_lib", "ops", "model", "train", "parallel", "build", "inference", "checkpoint", "metrics")


def __getattr__(name):
    if name in _LAZY:
        return importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
