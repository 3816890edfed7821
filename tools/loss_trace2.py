import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
import vptr_amd.ops as ops
from vptr_amd.train import NARTrainer
dev = torch.device("cuda:0"); B = 16
enc, dec, T = bench.build_models(dev, float(os.environ.get("DROPOUT", "0.1")))
tr = NARTrainer(enc, dec, T, batch_size=B, dec_weight_grads=bool(int(os.environ.get("DW", "1"))))
past, fut = bench.synth_batch(B, 0, dev)
if int(os.environ.get("GRAPH", "1")): tr.capture(past, fut, warmup=2)
for i in range(8):
    o = tr.step(past, fut)
    print("step", i, {k: round(float(v), 4) for k, v in o.items()})
