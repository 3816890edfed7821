#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the bench step's kernels (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_traffic -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_traffic -o w --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python - <<'PY'
import csv, collections, json
res = {}
for tag, cname in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open("gpurun_out/pmc_traffic/%s_counter_collection.csv" % tag)):
        if r["Counter_Name"] != cname: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res.setdefault(k, {})[cname] = v / n
        res[k]["launches"] = n
out = {k: v for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) * kv[1].get("launches", 0))) if "vptr_gemm" in k or v.get("FETCH_SIZE", 0) * v.get("launches", 0) > 1e5}
json.dump(out, open("gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
for k, v in list(out.items())[:14]:
    print("%-46s launches %5d  FETCH_SIZE %12.1f KB  WRITE_SIZE %12.1f KB per launch" % (k[:46], v.get("launches", 0), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
