#!/bin/bash
# does bench.py's headline pass ever hang? N runs per setting, each under its own timeout, one line per run (scratch, round 3)
N=${1:-15}
for t in 1 0; do
  for i in $(seq 1 $N); do
    s=$(date +%s.%N)
    GPR_LAUNCH_THREADS=$t timeout -k 5 60 python bench.py --no-extras --no-cpu-baseline --steps 1000 --min-seconds 0.3 > /tmp/hh.json 2> /tmp/hh.err
    rc=$?
    e=$(date +%s.%N)
    v=$(python -c "import json,sys; d=json.loads([l for l in open('/tmp/hh.json') if l.startswith('{')][-1]); print('%.3e us %.2f' % (d['value'], d['roofline']['kernel_us']))" 2>/dev/null)
    echo "threads=$t run=$i rc=$rc secs=$(python -c "print('%.1f' % ($e-$s))") $v" | tee -a gpurun_out/r3_hang_hunt2.log
    if [ $rc -ne 0 ]; then tail -3 /tmp/hh.err | tee -a gpurun_out/r3_hang_hunt2.log; fi
  done
done
