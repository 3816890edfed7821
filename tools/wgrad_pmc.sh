#!/bin/bash
# L2 / fabric counters of the grouped weight-gradient launch ALONE (tools/wgrad_standalone.py), one rocprofv3 --pmc pass per counter set
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/wgrad_pmc; rm -rf $O; mkdir -p $O
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-20)
  rocprofv3 --kernel-trace --pmc $c -d $O -o $t --output-format csv -- python tools/wgrad_standalone.py --order shape --reps 3 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("gpurun_out/wgrad_pmc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wgrad_p16" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c, v in sorted(agg.items()):
    print("%-16s %16.0f per launch (%d launches)" % (c, v / cnt[c], cnt[c]))
if "TCC_HIT_sum" in agg:
    print("L2 hit rate %.3f" % (agg["TCC_HIT_sum"] / (agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"])))
if "FETCH_SIZE" in agg:
    print("HBM-side GB per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: %.2f" % ((2 * agg["FETCH_SIZE"] / cnt["FETCH_SIZE"] + agg.get("WRITE_SIZE", 0) / max(cnt["WRITE_SIZE"], 1)) * 1024 / 1e9))
PY
