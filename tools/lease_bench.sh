#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 900 python bench.py > gpurun_out/$R/bench_final.log 2>gpurun_out/$R/bench_final.err; tail -1 gpurun_out/$R/bench_final.log > gpurun_out/$R/bench_final.json
for c in bair_far kth128; do timeout 600 python bench.py --config $c --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$R/bench_final_$c.json; done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_final.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["launch"])
print({k: (v.get("ms_per_step"), v.get("step_tflops")) for k, v in d["other_configs"].items()})
for c in ("bair_far", "kth128"):
    e = json.load(open("gpurun_out/r05/bench_final_%s.json" % c)); print(c, e["value"], e["ms_per_step"], e["config"]["step_tflops_per_gpu"], e["roofline"]["kernel"], e["roofline"]["frac"])
PY
