#!/bin/bash
# SQ-level PMC profile of the kernels of any command (GPU box).  usage: tools/pmc_cmd.sh <tag> <kernel-substring> <command...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; export KSUB=$2; shift 2
rm -rf gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/pmc_$TAG -o a --output-format csv -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d gpurun_out/pmc_$TAG -o b --output-format csv -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmc_$TAG -o c --output-format csv -- "$@" > /dev/null 2>&1
export TAG
python - <<'PY'
import csv, glob, collections, os
sub = os.environ["KSUB"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % os.environ["TAG"])):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
if os.environ.get("COMPACT"):
    print("%-40s %9s %7s %7s %7s %7s %7s" % ("kernel", "wave-cyc", "active", "valu", "lds", "wait", "w-inst"))
    rows = []
    for k, d in agg.items():
        g = lambda c: d.get(c, 0.0) / max(cnt[(k, c)], 1)
        wc = g("SQ_WAVE_CYCLES")
        if wc <= 0: continue
        rows.append((wc * cnt[(k, "SQ_WAVE_CYCLES")], k, wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc,
                     g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc))
    for r in sorted(rows, reverse=True)[:40]:
        print("%-40s %9.0f %7.2f %7.2f %7.2f %7.2f %7.2f" % r[1:])
else:
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()): print("    %-28s %14.0f per launch" % (c, v / cnt[(k, c)]))
PY
