#!/bin/bash
# round 6, lease 10: the round's evidence -- full GPU suite, profiles (kernel stats, PMC traffic, MFMA utilisation, exchange path), default bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=.
echo "### full GPU suite" > gpurun_out/r06_lease10.log
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 >> gpurun_out/r06_lease10.log
R=r06 bash tools/prof_round.sh > gpurun_out/r06_prof_round.log 2>&1
tail -40 gpurun_out/r06_prof_round.log >> gpurun_out/r06_lease10.log
# keep the summaries, drop the raw traces (gpurun merges at most 64 MiB back)
rm -rf gpurun_out/r06/stats gpurun_out/r06/pmc gpurun_out/r06/pmcx gpurun_out/r06/rccl gpurun_out/kstats
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -4 gpurun_out/r06_bench_default.err >> gpurun_out/r06_lease10.log
du -sh gpurun_out >> gpurun_out/r06_lease10.log
tail -30 gpurun_out/r06_lease10.log
