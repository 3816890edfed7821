#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/$R/gputests_1.log; cat gpurun_out/$R/gputests_1.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/$R/bench_f.log 2>gpurun_out/$R/bench_f.err; tail -1 gpurun_out/$R/bench_f.log > gpurun_out/$R/bench_f.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_f.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["launch"])
print(json.dumps(d.get("other_configs"), indent=0))
r = d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["all_gemm"])
PY
