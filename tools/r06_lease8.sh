#!/bin/bash
# round 6, lease 8: elimination runs of the fused norm1 + GELU + depthwise forward; full GPU suite; 
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_dwn_probe.log && : > $O
export PYTHONPATH=.
timeout 200 python tools/dwn_probe.py >> $O 2>&1
for v in "VPTR_DWN_LDS=1" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=1" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=2" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=4" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=16" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=32" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=3" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=55"; do
  env $v timeout 200 python tools/dwn_probe.py 2>&1 | grep "^env" >> $O
done
cat $O
echo "### full GPU suite" > gpurun_out/r06_lease8.log
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 >> gpurun_out/r06_lease8.log
tail -15 gpurun_out/r06_lease8.log
