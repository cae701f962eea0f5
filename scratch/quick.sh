#!/bin/bash
# on the GPU box: tests + short bench summary
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$?
grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | cut -c1-300
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench2.log 2>&1
tail -1 gpurun_out/bench2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g  kernel_us %.2f  sigma0_us %.2f  sigma0_value %.4g' % (d['value'], d['roofline']['kernel_us'], d['config']['sigma0_kernel_us'], d['config']['sigma0_env_steps_per_s']))"
