#!/bin/bash
# A/B of one environment knob on ONE box (box-to-box spread is +-2 ms): tools/ab_env.sh VAR valueA valueB [rounds]
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  for x in $A $B; do echo "$V=$x $(env $V=$x timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
done
