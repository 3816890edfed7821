#!/bin/bash
# L2 behaviour of the grouped weight-gradient launch inside the bench step (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_wgrad; mkdir -p gpurun_out/pmc_wgrad
SHORT="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs"
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_wgrad -o $t --output-format csv -- $SHORT > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_wgrad/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "wgrad" not in k and "gemm_p16" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for f in glob.glob("gpurun_out/pmc_wgrad/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]: dur[f].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f per launch" % (c, v / cnt[(k, c)]))
print("wgrad ms:", {f.split("/")[-1][:20]: [round(x, 2) for x in v] for f, v in dur.items()})
PY
