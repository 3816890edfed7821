#!/bin/bash
# round 6, lease 21: graph verification (verify_graph inside bench.py) of the BASELINE configs 2 / 4 / 5 with the round-6 kernels
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease21.log && : > $O
export PYTHONPATH=.
for c in mnist bair_far kth128; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['name'], d['ms_per_step'], d['config']['launch'], d['loss_sane'], json.dumps(d['graph_check'])[:400])" >> $O 2>&1
done
cat $O
