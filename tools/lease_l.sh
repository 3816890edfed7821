#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 1500 python -m pytest tests/test_01_p16_gpu.py tests/test_02_model_gpu.py tests/test_11_deterministic_gpu.py tests/test_20_graph_gpu.py tests/test_22_rccl_gpu.py -x -q 2>&1 | tail -6
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/$R/bench_l.log 2>gpurun_out/$R/bench_l.err; tail -1 gpurun_out/$R/bench_l.log > gpurun_out/$R/bench_l.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_l.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["launch"])
print(json.dumps(d.get("other_configs"), indent=0))
r = d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["all_gemm"]); print({k:v for k,v in r['per_kernel'].items() if 'wgrad' in k})
PY
for c in bair_far kth128; do timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', d['value'], d['ms_per_step'], d['config']['step_tflops_per_gpu'], {k:v for k,v in r['per_kernel'].items() if 'wgrad' in k})"; done
