#!/bin/bash
# Collects the judged profile artefacts on the GPU box: kernel-trace stats of bench.py and PMC passes (separate runs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.err
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmc_tcc.err
python - <<'PY'
import csv, collections, glob, os
O='gpurun_out/prof_round'
with open(O+'/summary.txt','w') as out:
    for f in sorted(glob.glob(O+'/trace/*/*_kernel_stats.csv')):
        out.write('== rocprofv3 --kernel-trace --stats -- python bench.py --steps 500 --warmup 50 (kernel_stats.csv, gpr kernels) ==\n')
        for r in csv.DictReader(open(f)):
            if 'gpr::' in r['Name']:
                out.write('%s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s\n'%(r['Name'],r['Calls'],r['AverageNs'],r['MinNs'],r['MaxNs'],r['Percentage']))
    for d in ('pmc_sq','pmc_fetch','pmc_write','pmc_tcc'):
        for f in sorted(glob.glob(O+'/'+d+'/*/*_counter_collection.csv')):
            agg=collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                if 'k_step' in r['Kernel_Name']:
                    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            out.write('== rocprofv3 --pmc (%s), mean per k_step dispatch ==\n'%d)
            for k,v in agg.items():
                out.write(k+'\n')
                for c,vals in v.items():
                    out.write('   %-22s %.6g   (per wavefront: %.1f)\n'%(c,sum(vals)/len(vals),sum(vals)/len(vals)/4096))
print(open(O+'/summary.txt').read())
PY
