#!/bin/bash
# round 6, lease 13: final state -- smoke(), two consecutive full -x suite runs, default bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease13.log && : > $O
export PYTHONPATH=.
echo "### smoke" >> $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> $O
for i in 1 2; do
  echo "### full GPU suite, run $i" >> $O
  timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 >> $O
done
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -4 gpurun_out/r06_bench_default.err >> $O
python - >> $O <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_default.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["config"]["launch"], d["config"].get("frozen_encoder_convs"))
print({k: v.get("ms_per_step") for k, v in d["other_configs"].items()})
print(d["roofline"]["traffic"], d["roofline"]["traffic_source"][:120], d["roofline"]["step"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["wall_s"])
PY
tail -30 $O
