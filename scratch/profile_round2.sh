#!/bin/bash
# The judged profile artefacts of the round, collected on the GPU box (each counter set in its own run, as gpurun requires):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py` (the headline command)  -> per-kernel average duration
#   2. the same for scratch/prof_workloads.py (every other kernel on its BASELINE configuration)
#   3. PMC passes for both: SQ mix, FETCH_SIZE, WRITE_SIZE, TCC requests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round2; rm -rf $O; mkdir -p $O
B="bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python $B > $O/bench_under_rocprof.json 2> $O/trace_bench.err
echo trace_bench done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_all -- python scratch/prof_workloads.py > $O/trace_all.log 2> $O/trace_all.err
echo trace_all done
S="bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-extras --min-seconds 0"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $S > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $S > /dev/null 2> $O/pmc_write.err
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python $S > /dev/null 2> $O/pmc_tcc.err
echo pmc_bench done
export PROF_STEPS=30
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_all -- python scratch/prof_workloads.py > /dev/null 2> $O/pmc_sq_all.err
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_all -- python scratch/prof_workloads.py > /dev/null 2> $O/pmc_fetch_all.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_all -- python scratch/prof_workloads.py > /dev/null 2> $O/pmc_write_all.err
echo pmc_all done
python scratch/profile_round2_summary.py $O
