#!/bin/bash
# round 6, lease 25: the 2-rank self-launch of the K64 bench on one GPU, repeated, every rank dumping its stacks if it is still running after 240 s
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease25.log && : > $O
export PYTHONPATH=.
python -c "import torch; torch.zeros(1, device='cuda'); import time; time.sleep(100000)" &   # a third process holding a context, like the pytest parent
HOLD=$!
for i in $(seq 1 14); do
  echo "### run $i" >> $O
  t0=$(date +%s)
  VPTR_BENCH_HANG_DUMP_S=240 VPTR_BENCH_SHARE_GPU=1 VPTR_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --batch 4 > gpurun_out/sl_$i.out 2> gpurun_out/sl_$i.err
  rc=$?
  echo "rc $rc in $(( $(date +%s) - t0 )) s" >> $O
  if [ $rc -ne 0 ]; then grep -v "^\s*$" gpurun_out/sl_$i.err | tail -80 >> $O; fi
  rm -f gpurun_out/sl_$i.out gpurun_out/sl_$i.err
done
kill $HOLD
grep -c "rc 0" $O
tail -150 $O
