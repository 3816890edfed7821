"""Per-shape GEMM table of one NAR train step (GPU box): launches, total ms, TFLOP/s by (M, N, K, a_mode, b_mode)."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
import vptr_amd.ops as ops
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
def hooked(dref, st):
    d = dref._obj
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = real(dref, st)
    e1.record()
    epi = ("g" if d.act == 1 else "") + ("P" if d.Dpre else "") + ("d" if d.dropout_p > 0 else "") + ("r" if d.residual else "") + ("s" if d.rowscale else "") + ("o" if d.d_p16 else "")
    recs.append(((d.M, d.N, d.K * max(d.ksegs, 1), d.a_mode, d.b_mode, max(d.batch, 1), d.conv_KH if d.a_mode in (2, 3) else 0, d.conv_stride if d.a_mode in (2, 3) else 0,
                  d.conv_transposed if d.a_mode == 2 else 0, epi), e0, e1))
    return rc
lib.vptr_gemm = hooked
trainer.step(past, fut)
torch.cuda.synchronize()
lib.vptr_gemm = real
by = {}
for k, e0, e1 in recs:
    d = by.setdefault(k, [0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in by.values())
print("%-60s %5s %9s %8s %8s" % ("M N K am bm batch kh stride transposed epilogue(g=GELU P=Dpre d=dropout r=residual s=rowscale o=P16 out)", "n", "ms", "us/call", "TF/s"))
for k, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k[:3]
    print("%-60s %5d %9.3f %8.1f %8.1f" % (" ".join(str(x) for x in k), n, ms, ms * 1e3 / n, 2.0 * M * N * K * k[5] * n / ms / 1e9))
print("total non-grouped GEMM ms: %.3f" % tot)
