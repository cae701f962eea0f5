#!/bin/bash
# The judged profile artefacts of round 4, collected on the GPU box (each counter set in its own run, as gpurun requires):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py` (the headline command: 4 sub-batches of 16 384 envs on 4 streams)
#   2. the same for scratch/prof_workloads.py (every other kernel on its BASELINE configuration)
#   3. PMC passes: HBM traffic (FETCH_SIZE, WRITE_SIZE) and the SQ instruction mix of the step kernel at the launch sizes the bench line
#      quotes (16 384 envs per launch: the sub-batch; 65 536: one launch per step; f64 at 16 384), and of every kernel of prof_workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round4; rm -rf $O; mkdir -p $O
B="bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-extras --min-seconds 0.3 --regions 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python $B > $O/bench_under_rocprof.json 2> $O/trace_bench.err
echo trace_bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench_s1 -- python $B --sub-batches 1 > $O/bench_s1_under_rocprof.json 2> $O/trace_bench_s1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench_f64 -- python $B --precision f64 > $O/bench_f64_under_rocprof.json 2> $O/trace_bench_f64.err
echo trace_bench variants done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_all -- python scratch/prof_workloads.py > $O/trace_all.log 2> $O/trace_all.err
echo trace_all done
export GPR_NO_STREAM_PROBE=1   # (counter collection serialises the launches: keep the launch shapes of the unprofiled run)
export GPR_LAUNCH_THREADS=0     # (... and every launch on the calling thread: one f64 FETCH_SIZE pass with launch workers never came back)
S="--steps 40 --warmup 60 --no-cpu-baseline --no-extras --min-seconds 0 --regions 1"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
for cfg in "s4:--sub-batches 4" "s1:--sub-batches 1" "f64:--sub-batches 4 --precision f64"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$tag -- python bench.py $S $fl > /dev/null 2> $O/pmc_fetch_$tag.err
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$tag -- python bench.py $S $fl > /dev/null 2> $O/pmc_write_$tag.err
  timeout -k 10 200 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_$tag -- python bench.py $S $fl > /dev/null 2> $O/pmc_sq_$tag.err
  echo pmc_$tag done
done
export PROF_STEPS=30
timeout -k 10 500 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_all -- python scratch/prof_workloads.py > /dev/null 2> $O/pmc_sq_all.err
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_all -- python scratch/prof_workloads.py > /dev/null 2> $O/pmc_fetch_all.err
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_all -- python scratch/prof_workloads.py > /dev/null 2> $O/pmc_write_all.err
echo pmc_all done
python scratch/profile_round4_summary.py $O
# the raw traces of this script are hundreds of MB (gpurun merges at most 64 MiB back): keep the summary, the stats tables and the logs
mkdir -p $O/raw_stats
for d in $O/trace_*; do t=$(basename $d); for f in $d/*/*_kernel_stats.csv; do cp $f $O/raw_stats/${t}_kernel_stats.csv 2>/dev/null; done; done
rm -rf $O/trace_bench $O/trace_bench_s1 $O/trace_bench_f64 $O/trace_all $O/pmc_*
du -sh $O
