#!/bin/bash
# usage: pmc2.sh <sigma>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcC gpurun_out/pmcD
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcC -- python scratch/one.py $1 > gpurun_out/pmcC.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d gpurun_out/pmcD -- python scratch/one.py $1 > gpurun_out/pmcD.log 2>&1
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc[CD]/*/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_step' not in k: continue
        agg[k[:58]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k)
        for c,vals in v.items():
            print('   %-24s per-wave=%.1f'%(c,sum(vals)/len(vals)/4096))
PY
