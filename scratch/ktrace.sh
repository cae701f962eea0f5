#!/bin/bash
# usage: ktrace.sh <sigma> <ncyc>: true kernel durations from the kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python scratch/one.py $1 300 $2 > gpurun_out/kt.log 2>&1
python - <<'PY'
import csv, glob, numpy as np
t=glob.glob('gpurun_out/kt/*/*_kernel_trace.csv')[0]
d=[]; starts=[]
for r in csv.DictReader(open(t)):
    if 'k_step' in r['Kernel_Name']:
        d.append(int(r['End_Timestamp'])-int(r['Start_Timestamp'])); starts.append(int(r['Start_Timestamp']))
d=np.array(d[50:]); gaps=np.diff(np.array(starts[50:]))
print('k_step: n=%d mean %.2f us  median %.2f  min %.2f; start-to-start median %.2f us'%(len(d), d.mean()/1e3, np.median(d)/1e3, d.min()/1e3, np.median(gaps)/1e3))
PY
