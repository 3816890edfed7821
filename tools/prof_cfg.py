"""Run a few train steps of one of BASELINE.json's other configurations (for rocprofv3): python tools/prof_cfg.py <4|5> [steps]"""
import contextlib, io, os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.model as M
from vptr_amd.train import FARTrainer, NARTrainer
dev = torch.device("cuda:0")
cfg, steps = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 4


def rnd(*shape):
    return torch.rand(shape, device=dev) * 0.36 - 0.22


with contextlib.redirect_stdout(io.StringIO()):
    if cfg == 4:
        enc, dec = M.VPTREnc(3, 528, 3, "zero"), M.VPTRDec(3, 528, 3, "Tanh", "zero")
        M.init_weights(enc); M.init_weights(dec)
        T = M.VPTRFormerFAR(2, 28, 8, 8, 528, 8, 12, 0.1, 4, 4, True)
        tr = FARTrainer(enc.to(dev), dec.to(dev), T.to(dev))
        past, fut = rnd(16, 2, 3, 64, 64), rnd(16, 28, 3, 64, 64)
    else:
        enc, dec = M.VPTREnc(1, 528, 3, "reflect"), M.VPTRDec(1, 528, 3, "Tanh", "reflect")
        M.init_weights(enc); M.init_weights(dec)
        T = M.VPTRFormerNAR(10, 40, 16, 16, 528, 8, 4, 8, 0.1, 8, 4, False, True)
        tr = NARTrainer(enc.to(dev), dec.to(dev), T.to(dev), batch_size=2)
        past, fut = rnd(2, 10, 1, 128, 128), rnd(2, 40, 1, 128, 128)
for _ in range(steps):
    out = tr.step(past, fut)
torch.cuda.synchronize()
print("config", cfg, "loss", float(out["T_total"]))
