#!/bin/bash
# A/B timing of library builds in ONE GPU call (devices differ by several percent): scratch/ab.sh "<workloads>" libA.so libB.so ...
# (workloads of scratch/variants.py); each build is timed twice, interleaved. GPR_LIB_PATH selects the build.
W="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do GPR_LIB_PATH=$PWD/$lib timeout -k 10 200 python scratch/variants.py $W 2>/dev/null | tr '\n' ';'; echo; done
done
