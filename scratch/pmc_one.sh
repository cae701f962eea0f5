#!/bin/bash
# usage: pmc_one.sh "<one.py args>" tag -- SQ instruction counters of the step kernel for one configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_one_$2; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $O/a -- python ${SCRIPT:-scratch/one.py} $1 > /dev/null 2> $O/a.err
python - "$O" <<'PY'
import csv, collections, glob, sys
O=sys.argv[1]
for f in sorted(glob.glob(O+'/a/*/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name'] or 'k_push' in r['Kernel_Name']:
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        n=len(v['SQ_WAVES']); half=n//2   # second half of the dispatches (steady state)
        waves=sum(v['SQ_WAVES'][half:])/(n-half)
        print(k[:80], 'dispatches', n, ' per wavefront:', '  '.join('%s %.0f' % (c.replace('SQ_',''), sum(x[half:])/(n-half)/waves) for c,x in sorted(v.items())))
PY
