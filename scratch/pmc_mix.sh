#!/bin/bash
# instruction mix of the step kernel (PMC, two passes of 8 SQ counters), mean per k_step dispatch and per wavefront
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_mix; rm -rf $O; mkdir -p $O
ARGS="bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-extras --min-seconds 0 ${BENCH_ARGS:-}"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/a -- python $ARGS > /dev/null 2> $O/a.err
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/b -- python $ARGS > /dev/null 2> $O/b.err
python - <<'PY'
import csv, collections, glob
O='gpurun_out/pmc_mix'
for d in ('a','b'):
    for f in sorted(glob.glob(O+'/'+d+'/*/*_counter_collection.csv')):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'k_step' in r['Kernel_Name'] or 'k_push' in r['Kernel_Name']:
                agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            waves = (sum(v['SQ_WAVES'])/len(v['SQ_WAVES'])) if 'SQ_WAVES' in v else None
            print(k[:90], 'dispatches', len(next(iter(v.values()))))
            for c,vals in v.items():
                m=sum(vals)/len(vals)
                print('   %-26s %.6g' % (c,m), ('(per wavefront %.1f)' % (m/waves)) if waves else '')
PY
