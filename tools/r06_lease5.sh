#!/bin/bash
# round 6, lease 5: Winograd F(4x4, 3x3) for the frozen encoder -- op tests, model parity, step A/B, per-kernel table
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease5.log && : > $O
export PYTHONPATH=.
echo "### winograd op tests" >> $O
timeout 900 python -m pytest tests/test_06_winograd_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $O
echo "### model parity (test_02, test_05, test_20)" >> $O
timeout 1500 python -m pytest tests/test_02_model_gpu.py tests/test_05_config_steps_gpu.py tests/test_20_graph_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $O
for i in 1 2 3; do for v in 1 0; do
  echo "VPTR_ENC_WINOGRAD=$v $(VPTR_ENC_WINOGRAD=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### kstats (winograd on)" >> $O
bash tools/kstats.sh 30 >> $O 2>&1
tail -120 $O
