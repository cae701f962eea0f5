#!/bin/bash
# thin wavefronts: bench.py's timed region per small shape and wavefront budget (GPR_THIN_WAVES), one launch per step and four sub-batches
W="${1:-m4c2 m4s m8s m16s}"; S="${2:-1}"
for tw in ${3:-0 1024 2048 4096}; do
  if [ "$tw" = default ]; then echo -n "default budget       "; timeout -k 10 300 python scratch/ab_sub.py "$W" "$S" 2>&1 | tail -1; continue; fi
  echo -n "GPR_THIN_WAVES=$tw  "; GPR_THIN_WAVES=$tw timeout -k 10 300 python scratch/ab_sub.py "$W" "$S" 2>&1 | tail -1
done
