"""Where the script-style ("drop-in") K64 iteration spends its time: train_NAR.py:49-107 on this package's modules with stock
torch.optim.AdamW / clip_grad_norm_ / criterion classes (vptr_amd.train.script_style_nar_iter).  Sections are bracketed with events on
the current stream AND host clocks: a section whose host time exceeds its device time is launch-bound.

    python tools/dropin_prof.py [steps]        (rocprofv3 --kernel-trace --stats -- python tools/dropin_prof.py for the kernel table)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
import vptr_amd.model as M  # noqa: E402
from vptr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, TF = bench.PER_GPU_BATCH, 10
ops.unregister_flat_slabs()
enc, dec, T = bench.build_models(dev, 0.1)
enc, dec = enc.eval(), dec.eval()
opt = torch.optim.AdamW(T.parameters(), lr=1e-4)
mse, gdl = M.MSELoss(), M.GDL(alpha=1)
bp = M.BiPatchNCE(N, TF, 8, 8, 1.0).to(dev)
past, fut = bench.synth_batch(N, 0, dev)

names = ["encoder", "zero_grad", "T_forward", "dec_forward", "losses", "backward", "clip", "optimizer"]
dev_ms = {k: 0.0 for k in names}
host_ms = {k: 0.0 for k in names}


class section:
    def __init__(self, name, rec):
        self.name, self.rec = name, rec

    def __enter__(self):
        if self.rec:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
            self.t0 = time.perf_counter()

    def __exit__(self, *a):
        if self.rec:
            host_ms[self.name] += (time.perf_counter() - self.t0) * 1e3
            self.e1.record()
            pending.append((self.name, self.e0, self.e1))


pending = []


def one(rec):
    with section("encoder", rec):
        with torch.no_grad():
            pf, ff = enc(past), enc(fut)
    with section("zero_grad", rec):
        T.train()
        T.zero_grad(set_to_none=True)
        dec.zero_grad(set_to_none=True)
    with section("T_forward", rec):
        pred_f = T(pf)
    with section("dec_forward", rec):
        pred = dec(pred_f)
    with section("losses", rec):
        a = T.NCE_projector(pred_f.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
        b = T.NCE_projector(ff.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
        loss = gdl(fut, pred) + mse(pred, fut) + 0.1 * bp(F.normalize(b, p=2.0, dim=2), F.normalize(a, p=2.0, dim=2))
    with section("backward", rec):
        loss.backward()
    with section("clip", rec):
        torch.nn.utils.clip_grad_norm_(T.parameters(), max_norm=1.0, norm_type=2)
    with section("optimizer", rec):
        opt.step()


for _ in range(3):
    one(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one(False)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
t0 = time.perf_counter()
for _ in range(steps):
    one(True)
torch.cuda.synchronize()
wall_rec = (time.perf_counter() - t0) / steps * 1e3
for name, e0, e1 in pending:
    dev_ms[name] += e0.elapsed_time(e1)
print("script-style K64 iteration, batch %d: %.2f ms/step (%.2f with section events)" % (N, wall, wall_rec))
print("%-12s %10s %10s" % ("section", "device ms", "host ms"))
for k in names:
    print("%-12s %10.2f %10.2f" % (k, dev_ms[k] / steps, host_ms[k] / steps))
print("%-12s %10.2f %10.2f" % ("sum", sum(dev_ms.values()) / steps, sum(host_ms.values()) / steps))

if os.environ.get("VPTR_HOST_PROFILE") == "1":       # where the HOST time goes (the iteration is launch-bound): cProfile over 5 iterations
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        one(False)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(60)
