#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcA gpurun_out/pmcB
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcA -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmcA.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d gpurun_out/pmcB -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmcB.log 2>&1
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc[AB]/*/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_step' not in k: continue
        agg[k[:58]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k)
        for c,vals in v.items():
            print('   %-24s mean=%.4g per-wave=%.1f'%(c,sum(vals)/len(vals),sum(vals)/len(vals)/4096))
PY
