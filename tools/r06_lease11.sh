#!/bin/bash
# round 6, lease 11: cooperative one-pass LayerNorm((F,H,W)) backward -- op test, model parity, step A/B, kernel table
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease11.log && : > $O
export PYTHONPATH=.
echo "### op test" >> $O
timeout 600 python -m pytest tests/test_01_p16_gpu.py -q -m gpu -k "coop" 2>&1 | tail -12 >> $O
echo "### model parity" >> $O
timeout 1500 python -m pytest tests/test_00_ops_gpu.py tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py tests/test_05_config_steps_gpu.py tests/test_20_graph_gpu.py -q -m gpu 2>&1 | tail -8 >> $O
for i in 1 2 3; do for v in "VPTR_NORM_COOP=1" "VPTR_NORM_COOP=0"; do
  echo "$v $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### kstats" >> $O
bash tools/kstats.sh 24 >> $O 2>&1
rm -rf gpurun_out/kstats
tail -70 $O
