"""Checkpoint format of the reference (`utils/train_summary.py:130-160` save_ckpt / load_ckpt, `:10-38` resume_training),
so that the authors' released `epoch_N.tar` files and checkpoints written by the reference's trainers load into the
vptr_amd modules, and files written here load back into the reference.

Layout of the `torch.save`d dict: {'epoch': int, 'loss_dict': {name: Loss_tuple(train=[...], val=[...]), 'epochs': int},
'Module_state_dict': {module name: state_dict}, 'optimizer_state_dict': {optimizer name: state_dict}, 'code': {...}}.
`loss_dict` pickles instances of the reference class `utils.train_summary.Loss_tuple`; unpickling maps that global to
`LossTuple` below, and pickling writes it back under the reference's name.  The reference also embeds a copy of its source
tree under 'code'; this writer stores an empty dict there (readers ignore it).
"""
import contextlib
import pickle
import sys
import types
from collections import OrderedDict
from pathlib import Path

import torch


class LossTuple(object):
    """Per-loss history (utils/train_summary.py:92-95)."""

    def __init__(self):
        self.train = []
        self.val = []


_REF_MODULE, _REF_NAME = "utils.train_summary", "Loss_tuple"


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_MODULE and name == _REF_NAME:
            return LossTuple
        return super().find_class(module, name)


class _pickle_shim:
    """pickle-module stand-in for torch.load: identical to pickle except for the Loss_tuple mapping"""
    __name__ = "pickle"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps = staticmethod(pickle.dump), staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    UnpicklingError, PicklingError = pickle.UnpicklingError, pickle.PicklingError
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL


@contextlib.contextmanager
def _reference_namespace():
    """While saving, LossTuple pickles as utils.train_summary.Loss_tuple (pickle verifies that the name resolves)."""
    saved = {k: sys.modules.get(k) for k in ("utils", _REF_MODULE)}
    real = saved[_REF_MODULE] is not None and getattr(saved[_REF_MODULE], _REF_NAME, None) is not None
    old = (LossTuple.__module__, LossTuple.__qualname__, LossTuple.__name__)
    try:
        if not real:
            pkg = saved["utils"] if saved["utils"] is not None else types.ModuleType("utils")
            mod = types.ModuleType(_REF_MODULE)
            setattr(mod, _REF_NAME, LossTuple)
            sys.modules["utils"], sys.modules[_REF_MODULE] = pkg, mod
            LossTuple.__module__, LossTuple.__qualname__, LossTuple.__name__ = _REF_MODULE, _REF_NAME, _REF_NAME
        yield getattr(saved[_REF_MODULE], _REF_NAME) if real else None
    finally:
        LossTuple.__module__, LossTuple.__qualname__, LossTuple.__name__ = old
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def init_loss_dict(loss_name_list, history_loss_dict=None):
    """utils/train_summary.py:97-115"""
    loss_dict = {name: LossTuple() for name in loss_name_list}
    loss_dict["epochs"] = 0
    if history_loss_dict is not None:
        for k, v in history_loss_dict.items():
            loss_dict[k] = v
        for k in list(loss_dict):
            if k not in history_loss_dict:
                lt = LossTuple()
                lt.train = [0] * history_loss_dict["epochs"]
                lt.val = [0] * history_loss_dict["epochs"]
                loss_dict[k] = lt
    return loss_dict


def save_ckpt(Modules_dict, Optimizers_dict, epoch, loss_dict, save_dir):
    """utils/train_summary.py:130-148 -> save_dir/epoch_{epoch}.tar"""
    Path(save_dir).mkdir(parents=True, exist_ok=True)
    ckpt_file = Path(save_dir).joinpath("epoch_%d.tar" % epoch)
    # histories written by the reference itself carry its own class; everything else is converted to plain LossTuple
    payload = {
        "epoch": epoch,
        "loss_dict": loss_dict,
        "Module_state_dict": {k: m.state_dict() for k, m in Modules_dict.items()},
        "optimizer_state_dict": {k: m.state_dict() for k, m in Optimizers_dict.items()},
        "code": {},
    }
    with _reference_namespace() as ref_cls:
        if ref_cls is not None:  # the reference itself is importable in this process: hand its own class to pickle
            conv = {}
            for k, v in loss_dict.items():
                if isinstance(v, LossTuple):
                    r = ref_cls()
                    r.train, r.val = list(v.train), list(v.val)
                    v = r
                conv[k] = v
            payload["loss_dict"] = conv
        torch.save(payload, ckpt_file.absolute().as_posix())
    return ckpt_file


def load_ckpt(ckpt_file, map_location=None):
    """utils/train_summary.py:150-160 -> (Modules_state_dict, Optimizers_state_dict, epoch, loss_dict, code)"""
    ckpt = torch.load(ckpt_file, map_location=map_location, pickle_module=_pickle_shim, weights_only=False)
    return ckpt["Module_state_dict"], ckpt["optimizer_state_dict"], ckpt["epoch"], ckpt["loss_dict"], ckpt.get("code", {})


def _strip_module_prefix(state_dict):
    """checkpoints written under DistributedDataParallel prefix every key with `module.` (train_summary.py:16-21)"""
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())


def resume_training(module_dict, optimizer_dict, resume_ckpt, loss_name_list=None, map_location=None):
    """utils/train_summary.py:10-38: load modules + optimizers from a checkpoint; returns (loss_dict, start_epoch), or
    (start_epoch, history_loss_dict) when map_location is given -- the reference's (asymmetric) convention."""
    modules_sd, optims_sd, start_epoch, history_loss_dict, _ = load_ckpt(resume_ckpt, map_location)
    for k, m in module_dict.items():
        sd = modules_sd[k]
        try:
            m.load_state_dict(sd)
        except RuntimeError:
            m.load_state_dict(_strip_module_prefix(sd))
    for k, o in optimizer_dict.items():
        o.load_state_dict(optims_sd[k])
    if map_location is None:
        return init_loss_dict(loss_name_list, history_loss_dict), start_epoch
    return start_epoch, history_loss_dict
