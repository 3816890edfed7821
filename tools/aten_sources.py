"""Which Python lines of the NAR train step launch ATen (non-vptr) kernels: torch.profiler with stacks, grouped by the innermost
repo frame.  GPU box."""
import os, sys, collections
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from vptr_amd.train import NARTrainer
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
enc, dec, tr = bench.build_models(dev, 0.1)
trainer = NARTrainer(enc, dec, tr, batch_size=16, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
past, fut = bench.synth_batch(16, 0, dev)
for _ in range(3):
    trainer.step(past, fut)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    trainer.step(past, fut)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    dt = ev.self_device_time_total if hasattr(ev, "self_device_time_total") else ev.self_cuda_time_total
    if dt <= 0:
        continue
    frame = "?"
    for f in ev.stack or []:
        if "/vptr_amd/" in f or "bench.py" in f:
            frame = f.split("/root/repo/")[-1] if "/root/repo/" in f else f[-90:]
            break
    k = (ev.name, frame + "  " + str([tuple(x) for x in (ev.input_shapes or []) if x])[:80])
    by[k][0] += 1
    by[k][1] += dt
tot = sum(v[1] for v in by.values())
print("ATen ops with device time: %.3f ms / step" % (tot / 1e3))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%4d  %8.1f us  %-28s %s" % (v[0], v[1], k[0], k[1]))
