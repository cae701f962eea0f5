#!/bin/bash
# usage: pmc3.sh <sigma> <ncyc> : instruction counts of the step kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcE
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcE -- python scratch/one.py $1 60 $2 > gpurun_out/pmcE.log 2>&1
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmcE/*/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_step' not in k: continue
        agg[k[:58]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, ' '.join('%s=%.0f'%(c,sum(vals[20:])/len(vals[20:])/4096) for c,vals in sorted(v.items())))
PY
