"""Which aten ops (with shapes) run inside one train step: finds stray copies / fills / adds (GPU box)."""
import os, sys, collections
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from vptr_amd.train import NARTrainer
dev = torch.device("cuda:0")
enc, dec, T = bench.build_models(dev, 0.1)
tr = NARTrainer(enc, dec, T, batch_size=16)
past, fut = bench.synth_batch(16, 0, dev)
for _ in range(3):
    tr.step(past, fut)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(past, fut)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0.0)
    if e.key.startswith("aten::") and t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:100]))
for t, c, n, sh in sorted(rows, reverse=True)[:40]:
    print("%-26s %4d x %9.1f us total  %s" % (n, c, t, sh))
