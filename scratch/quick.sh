#!/bin/bash
# on the GPU box: tests + short bench summary
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$?
grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | cut -c1-300
timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench2.log 2>&1; echo bench_exit=$?
tail -1 gpurun_out/bench2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('value %.4g  kernel_us %.2f  sigma0_us %.2f  done_frac %.3f' % (d['value'], d['roofline']['kernel_us'], c.get('sigma0_kernel_us', -1), c['done_fraction_per_step']))
for k in ('other_batch_sizes','reference_dtype'):
    for n,v in c.get(k,{}).items(): print('  %s %s: %.2f us  %.4g env-steps/s  hbm %.3f' % (k, n, v['kernel_us'], v['env_steps_per_s'], v['hbm_frac']))
for n,v in c.get('rollout_launch',{}).items(): print('  rollout_launch %s: %.2f us/step  %.4g env-steps/s' % (n, v['us_per_step'], v['env_steps_per_s']))
"
