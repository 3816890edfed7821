"""TEST INFRASTRUCTURE -- container-only helper that imports the real reference.

Imports XiYe20/VPTR from /root/reference on CPU so that (a) the oracle
restatement in `oracle/vptr_oracle.py` can be validated against it and (b)
golden fixtures can be generated (`oracle/make_golden.py`).  /root/reference
does not exist on the GPU box: nothing under tests -m gpu, smoke() or bench.py
may import this module.

The reference needs three third-party modules that are not installed here
(timm, torchvision, cv2); only two trivial helpers of timm are used on the
model path (`to_2tuple`, `trunc_normal_`), torchvision/cv2 only by the data
loaders that `utils/__init__.py` drags in.  We register empty stand-in modules
for the import to succeed; no reference source is copied.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def import_reference():
    """Returns the reference's `model` package (CPU)."""
    sys.dont_write_bytecode = True
    if "model" in sys.modules and getattr(sys.modules["model"], "__file__", "").startswith(REFERENCE_ROOT):
        return sys.modules["model"]

    def mk(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "timm" not in sys.modules:
        timm, tm, tl = mk("timm"), mk("timm.models"), mk("timm.models.layers")
        timm.models, tm.layers = tm, tl
        tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        tl.trunc_normal_ = torch.nn.init.trunc_normal_
    if "torchvision" not in sys.modules:
        tv, tvu, tvt = mk("torchvision"), mk("torchvision.utils"), mk("torchvision.transforms")
        tv.utils, tv.transforms = tvu, tvt
        for n in ["Compose", "Normalize", "ToTensor", "ToPILImage", "Resize", "CenterCrop", "Pad"]:
            setattr(tvt, n, type(n, (), {"__init__": lambda s, *a, **k: None}))
        tvt.functional = types.SimpleNamespace()
    if "cv2" not in sys.modules:
        mk("cv2")

    # our own repo also has a top-level `model` shim; make sure the reference wins here
    saved = [p for p in sys.path]
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import model  # noqa: the reference's package
        import utils.position_encoding as pe
    finally:
        sys.path[:] = saved
    cpu = torch.device("cpu")
    pe.PositionEmbeddding2D.__init__.__defaults__ = (10000, False, None, cpu)
    pe.PositionEmbeddding3D.__init__.__defaults__ = (10000, False, None, cpu)
    return model
