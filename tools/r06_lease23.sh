#!/bin/bash
# round 6, lease 23: last full -x suite run of the committed state
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease23.log && : > $O
export PYTHONPATH=.
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 >> $O
cat $O
