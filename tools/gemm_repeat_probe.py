"""In-step context probe (GPU box): every plain vptr_gemm launch of one train step is issued TWICE (idempotent: D is
rewritten with the same values; atomic / accumulating launches are skipped) and both are event-timed.  If the repeat is
much faster than the first issue, the in-step slowdown vs the micro-benchmark is a cold-start effect (caches, I$),
otherwise it is tied to the launch itself (addresses, epilogue, data)."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from vptr_amd._lib import lib
from vptr_amd.train import NARTrainer

dev = torch.device("cuda:0")
enc, dec, tr = bench.build_models(dev, 0.1)
trainer = NARTrainer(enc, dec, tr, batch_size=16, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
past, fut = bench.synth_batch(16, 0, dev)
for _ in range(3):
    trainer.step(past, fut)
torch.cuda.synchronize()
recs = []
real = lib.vptr_gemm
def ev():
    return torch.cuda.Event(enable_timing=True)
def hooked(dref, st):
    d = dref._obj
    if d.atomic or d.split_k > 1:
        return real(dref, st)
    e = [ev() for _ in range(4)]
    e[0].record(); rc = real(dref, st); e[1].record()
    e[2].record(); real(dref, st); e[3].record()
    feat = "res" if d.residual else ""
    feat += "+drop" if d.dropout_p > 0 else ""
    feat += "+rs" if d.rowscale else ""
    feat += "+act%d" % d.act if d.act else ""
    feat += "+pre" if d.Dpre else ""
    recs.append(((d.M, d.N, d.K * max(d.ksegs, 1), d.a_mode, d.b_mode, max(d.batch, 1), feat), e))
    return rc
lib.vptr_gemm = hooked
trainer.step(past, fut)
torch.cuda.synchronize()
lib.vptr_gemm = real
by = {}
for k, e in recs:
    d = by.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += e[0].elapsed_time(e[1]); d[2] += e[2].elapsed_time(e[3])
print("%-50s %4s %9s %9s" % ("M N K am bm batch epilogue", "n", "first us", "repeat us"))
for k, (n, a, b) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %4d %9.1f %9.1f" % (" ".join(str(x) for x in k), n, a * 1e3 / n, b * 1e3 / n))
