"""Does the panel-synchronous weight-gradient launch ever time out?  Runs N eager train steps (the launch follows the backward pass's
last kernels, as in the profiled step) and reads the per-XCD timeout counts (vptr_wgrad_sync_stats); also prints the launch durations (HIP events around the flush).
    python tools/wgrad_sync_probe.py [steps]"""
import ctypes
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
from vptr_amd import _lib, ops  # noqa: E402
from vptr_amd.train import NARTrainer  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
enc, dec, T = bench.build_models(dev, 0.1)
tr = NARTrainer(enc, dec, T, batch_size=16, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
past, fut = bench.synth_batch(16, 0, dev)
recs = []
ops.profiling.gemm = recs
for _ in range(steps):
    tr.step(past, fut)
torch.cuda.synchronize()
ops.profiling.gemm = None
ms = [e0.elapsed_time(e1) for key, fl, e0, e1 in recs if len(key) > 4 and str(key[4]).startswith("grouped") and key[4] != "grouped_split"]
print("grouped weight-gradient launches: %d, ms: %s" % (len(ms), " ".join("%.2f" % m for m in ms)))
out = torch.zeros(8, dtype=torch.int32, device=dev)
from vptr_amd._lib import check, ptr, stream  # noqa: E402
check(_lib.lib.vptr_wgrad_sync_stats(ptr(out), stream()), "vptr_wgrad_sync_stats")
print("timeouts per XCD (cumulative):", out.tolist())
