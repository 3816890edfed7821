"""Stability soak of the NAR train step (GPU box): 400 steps, device memory and loss trajectory."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from vptr_amd.train import NARTrainer
dev = torch.device("cuda:0")
enc, dec, T = bench.build_models(dev, 0.1)
tr = NARTrainer(enc, dec, T, batch_size=16)
past, fut = bench.synth_batch(16, 0, dev)
t0 = time.perf_counter()
for i in range(400):
    out = tr.step(past, fut)
    if i in (9, 99, 199, 399):
        torch.cuda.synchronize()
        print("step %3d  loss %.5f  grad_norm %.4f  allocated %.2f GB  reserved %.2f GB  peak %.2f GB  %.1f ms/step" % (
            i + 1, float(out["T_total"]), float(out["grad_norm"]), torch.cuda.memory_allocated() / 2**30,
            torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30, (time.perf_counter() - t0) / (i + 1) * 1e3))
