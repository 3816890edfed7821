#!/bin/bash
# functional check of the data-parallel step on a ONE-GPU box: 2 ranks on cuda:0, gradient exchange through gloo, with
# and without the overlapped chunked exchange; the final losses must agree.  Runs bench.py with its DEFAULT instrumented roofline
# pass, which rank 0 executes alone: it must not issue a collective (the other ranks wait at the final barrier).
for o in 1 0; do
  VPTR_DP_OVERLAP=$o VPTR_BENCH_SHARE_GPU=1 VPTR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29510 + o)) bench.py --gpus 2 --steps 4 --warmup 2 --batch 4 \
    --dropout 0 --no-other-configs 2>/dev/null | tail -1 > /tmp/dp_$o.json
  python - <<PY
import json
d = json.load(open("/tmp/dp_$o.json"))
print("overlap=$o  n_gpus", d["n_gpus"], " ms/step", d["ms_per_step"], " final_loss", d["final_loss"])
PY
done
