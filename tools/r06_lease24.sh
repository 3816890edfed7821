#!/bin/bash
# round 6, lease 24: vectorised AdamW kernel -- optimizer / step tests, kernel table, full suite, default bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease24.log && : > $O
export PYTHONPATH=.
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 >> $O
bash tools/kstats.sh 30 2>&1 | grep -i "kernel time\|adamw\|weight_planes\|sumsq" >> $O
rm -rf gpurun_out/kstats
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python - >> $O <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_default.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: v.get("ms_per_step") for k, v in d["other_configs"].items()})
print(d["roofline"]["hbm_side"])
PY
cat $O
