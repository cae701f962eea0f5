#!/bin/bash
# A/B timing of step-kernel builds in ONE GPU call (devices differ by several percent): scratch/ab.sh libA.so libB.so ...
# each lib is timed 3 times, interleaved; prints kernel_us per run
for rep in 1 2 3; do
  for lib in "$@"; do
    GPR_LIB_PATH=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 1000 --warmup 100 ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'rep $rep', 'kernel_us %.2f' % d['roofline']['kernel_us'])"
  done
done
