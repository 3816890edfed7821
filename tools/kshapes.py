"""Per-(kernel, grid) durations from a rocprofv3 kernel trace (tools/kstats.sh writes gpurun_out/kstats/b_kernel_trace.csv):
python tools/kshapes.py [trace.csv] [steps] [filter,filter,...]"""
import collections, csv, sys
f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kstats/b_kernel_trace.csv"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
keys = sys.argv[3].split(",") if len(sys.argv) > 3 else ["norm_act", "dwconv", "ln_", "act_bwd4", "rowmod", "colstats", "attn16", "partial_reduce", "conv7", "bnrelu"]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in keys):
        by[(n[:44], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v.sort()
    print("%-46s grid %6d x %4d  n/step %6.1f  med %7.1f us  total/step %7.3f ms" % (k[0], k[1], k[2], len(v) / steps, v[len(v) // 2], sum(v) / steps / 1e3))
