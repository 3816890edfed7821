#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/prof_round.sh > gpurun_out/prof_round_r05.log 2>&1; head -14 gpurun_out/$R/r05_bench_kernel_stats.md
timeout 900 python bench.py > gpurun_out/$R/bench_final.log 2>gpurun_out/$R/bench_final.err; tail -1 gpurun_out/$R/bench_final.log > gpurun_out/$R/bench_final.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_final.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["launch"])
print({k: (v.get("ms_per_step"), v.get("step_tflops")) for k, v in d["other_configs"].items()})
r = d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["traffic"], str(r["traffic_source"])[:80]); print(r["operand_stream"]["frac"], r["hbm_side"]["frac"], d["cpu_baseline"]["value"])
PY
