#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/wgrad_sync_probe.py 60 2>&1 | grep -v amdgpu | cut -c1-400
timeout 600 python tools/soak.py 2>&1 | grep -v amdgpu
timeout 600 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from vptr_amd.train import NARTrainer
from vptr_amd import _lib
from vptr_amd._lib import check, ptr, stream
dev = torch.device("cuda:0")
enc, dec, T = bench.build_models(dev, 0.1)
tr = NARTrainer(enc, dec, T, batch_size=16)
past, fut = bench.synth_batch(16, 0, dev)
tr.capture(past, fut, warmup=3)
for i in range(600):
    out = tr.step(past, fut)
torch.cuda.synchronize()
st = torch.zeros(8, dtype=torch.int32, device=dev)
check(_lib.lib.vptr_wgrad_sync_stats(ptr(st), stream()), "stats")
print("600 graph replays: loss %.5f grad_norm %.4f  sync timeouts per XCD %s" % (float(out["T_total"]), float(out["grad_norm"]), st.tolist()))
PY
