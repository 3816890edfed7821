"""Timing of the other steps of the pipeline on one MI355X (not bench lines; DESIGN.md section 6 quotes them):
stage-1 AE+GAN step, stage-2 FAR step, NAR / FAR inference rollouts.  Synthetic inputs resident in HBM."""
import os, sys, time
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.model as M
from vptr_amd.train import AETrainer, FARTrainer
from vptr_amd.inference import nar_rollout, far_rollout

dev = torch.device("cuda:0")


def timeit(fn, warm=3, steps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def frames(n, t, c=1, seed=0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(0, 1, size=(n, t, c, 64, 64)).astype(np.float32)).to(dev)


torch.manual_seed(0)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    enc = M.VPTREnc(1, 528, 3).to(dev); dec = M.VPTRDec(1, 528, 3, out_layer="Sigmoid").to(dev)
    disc = M.VPTRDisc(1, ndf=64, n_layers=3).to(dev)
    M.init_weights(enc); M.init_weights(dec); M.init_weights(disc)
# stage 1: train_AutoEncoder.py (N = 32 clips of 10+10 frames in the script; 8 here and in the script's MNIST variant 4)
for n in (4, 8):
    tr = AETrainer(enc, dec, disc, lr=2e-4, lam_gan=0.01)
    p, f = frames(n, 10, seed=1), frames(n, 10, seed=2)
    ms = timeit(lambda: tr.step(p, f))
    print("stage-1 AE+GAN step   batch %2d x 20 frames: %8.2f ms/step  %8.1f frames/s" % (n, ms, n * 20 / ms * 1e3))
# stage 2 FAR (train_FAR.py: 12 layers, rpe False, N = 10, T = 19)
far = M.VPTRFormerFAR(10, 10, 8, 8, 528, 8, 12, 0.1, 4, 4, False).to(dev)
enc.eval(); dec.eval()
trf = FARTrainer(enc, dec, far, lr=1e-4)
p, f = frames(10, 10, seed=3), frames(10, 10, seed=4)
ms = timeit(lambda: trf.step(p, f))
print("stage-2 FAR step      batch 10, T = 19 (12 layers): %8.2f ms/step  %8.1f predicted frames/s" % (ms, 10 * 19 / ms * 1e3))
# inference
nar = M.VPTRFormerNAR(10, 10, 8, 8, 528, 8, 4, 8, 0.1, 4, 4, False, True).to(dev)
p = frames(16, 10, seed=5)
ms = timeit(lambda: nar_rollout(enc, dec, nar, p, rounds=1))
print("NAR inference 10->10  batch 16: %8.2f ms  %8.1f predicted frames/s" % (ms, 160 / ms * 1e3))
ms = timeit(lambda: far_rollout(enc, dec, far, p[:8], 10), warm=1, steps=3)
print("FAR rollout 10->10    batch  8 (10 autoregressive passes with Dec->Enc re-encoding): %8.2f ms  %8.1f predicted frames/s" % (ms, 80 / ms * 1e3))
