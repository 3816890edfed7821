"""cProfile of the host side of one NAR train step at a host-bound batch size (GPU box)."""
import cProfile, pstats, os, sys, io
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from vptr_amd.train import NARTrainer

dev = torch.device("cuda:0")
enc, dec, tr = bench.build_models(dev, 0.1)
B = int(os.environ.get("BATCH", 4))
trainer = NARTrainer(enc, dec, tr, batch_size=B, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
past, fut = bench.synth_batch(B, 0, dev)
for _ in range(5):
    trainer.step(past, fut)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    trainer.step(past, fut)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:44]))
