#!/bin/bash
# round-5 lease A: the self-launch tests, the default bench line, config 4 / 5 kernel tables
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 1500 python -m pytest tests/test_24_bench_launch_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/$R/test24.log; cat gpurun_out/$R/test24.log
timeout 900 python bench.py > gpurun_out/$R/bench_default.log 2>gpurun_out/$R/bench_default.err; tail -1 gpurun_out/$R/bench_default.log > gpurun_out/$R/bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["launch"])
print(json.dumps(d.get("other_configs"), indent=0))
r = d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["all_gemm"]); print(json.dumps(r["hbm_side"])); print(json.dumps(r["infinity_cache_side"]))
print(d.get("cpu_baseline", {}).get("value"))
PY
for c in bair_far kth128; do timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/$R/bench_$c.json 2>gpurun_out/$R/bench_$c.err; python -c "
import json,sys
d=json.loads(open('gpurun_out/$R/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['step_tflops_per_gpu'], d['roofline']['kernel'], d['roofline']['frac'])"; done
bash tools/prof_cfg.sh 4 pmc | head -50
bash tools/prof_cfg.sh 5 pmc | head -50
