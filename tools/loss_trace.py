import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
import vptr_amd.ops as ops
from vptr_amd.train import NARTrainer
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 16))
for graph in (0, 1):
    for dw in (True, False):
        enc, dec, T = bench.build_models(dev, 0.1)
        tr = NARTrainer(enc, dec, T, batch_size=B, dec_weight_grads=dw)
        past, fut = bench.synth_batch(B, 0, dev)
        if graph: tr.capture(past, fut, warmup=2)
        ls = []
        for i in range(16):
            o = tr.step(past, fut); ls.append(round(float(o["T_total"]), 4))
        print("graph", graph, "dec_wgrad", dw, ls, "gn", float(o["grad_norm"]))
