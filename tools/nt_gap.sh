#!/bin/bash
# where do the ~9 us between the probe's nt kernel and the product's go?  GPU-side durations of both under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $R/tools/gemm_standalone.py > /dev/null 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob('/tmp/pp/**/p_kernel_trace.csv', recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_p16' in r['Kernel_Name']:
        by[int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(by.items()):
    v.sort(); print('product tiles', k, len(v), 'min %.1f med %.1f max %.1f' % (v[0], v[len(v)//2], v[-1]))
PY
ROT=6 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q --output-format csv -- $R/tools/_bin/gemm_p16_probe nt > /dev/null 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob('/tmp/pq/**/q_kernel_trace.csv', recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_nt' in r['Kernel_Name']:
        by[(r['Kernel_Name'][:40], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(by.items()):
    v.sort(); print('probe', k, len(v), 'min %.1f med %.1f max %.1f' % (v[0], v[len(v)//2], v[-1]))
PY
