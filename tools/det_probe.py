"""Which gradients are NOT reproducible?  Two runs of one train step from the same state under ops.set_deterministic(True); lists the
parameters whose gradient ranges differ (name, differing elements, max |d|) -- the work list for the deterministic mode.
    python tools/det_probe.py [tiny_nar|tiny_far|k64|k64_n16]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from vptr_amd import ops  # noqa: E402
import test_11_deterministic_gpu as T11  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "tiny_nar"
ops.manual_seed(dev, 77)
ops.set_deterministic(os.environ.get("DET", "1") == "1")
tr, past, fut = (T11._k64(dev, 16 if which == "k64_n16" else 2) if which.startswith("k64") else T11._tiny(dev, which == "tiny_far"))
snap = tr._snapshot()
runs = []
for r in range(2):
    tr._restore(snap)
    out = tr.step(past, fut)
    runs.append(({k: float(v) for k, v in out.items()}, tr.opt.grad.clone(), tr.opt.flat.clone()))
print(which, "terms equal:", runs[0][0] == runs[1][0], runs[0][0], runs[1][0])
ga, gb = runs[0][1], runs[1][1]
print("grad slab differing elements: %d of %d" % (int((ga != gb).sum()), ga.numel()))
names = [n for n, p in tr.T.named_parameters() if p.requires_grad]
for (off, n, _), name in zip(tr.opt._layout, names):
    a, b = ga[off:off + n], gb[off:off + n]
    d = int((a != b).sum())
    if d:
        print("  %-70s %8d / %-8d max|d| %.3e  |g| %.3e" % (name, d, n, float((a - b).abs().max()), float(a.norm())))
