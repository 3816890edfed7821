#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/$R/gputests_final.log; cat gpurun_out/$R/gputests_final.log
timeout 900 python bench.py > gpurun_out/$R/bench_final.log 2>gpurun_out/$R/bench_final.err; tail -1 gpurun_out/$R/bench_final.log > gpurun_out/$R/bench_final.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_final.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["launch"])
print(json.dumps(d.get("other_configs"), indent=0))
r = d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["all_gemm"], r["operand_stream"]); print(r["hbm_side"]["achieved"], r["hbm_side"]["frac"])
print(d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
bash tools/prof_round.sh > gpurun_out/prof_round_r05.log 2>&1; head -12 gpurun_out/$R/r05_bench_kernel_stats.md
bash tools/prof_cfg.sh 4 pmc > gpurun_out/$R/prof_cfg4.log 2>&1; head -14 gpurun_out/$R/r05_cfg4_kernel_stats.md | tail -8
bash tools/prof_cfg.sh 5 pmc > gpurun_out/$R/prof_cfg5.log 2>&1; head -14 gpurun_out/$R/r05_cfg5_kernel_stats.md | tail -8
