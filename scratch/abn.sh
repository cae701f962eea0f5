#!/bin/bash
# interleaved timing of the in-tree library and any number of builds under scratch/libs: abn.sh "<workloads>" "<S values>" a.so b.so ...
W="$1"; S="$2"; shift 2
for rep in 1 2; do
  for lib in gymnasium-planar-robotics_amd/libgpr_hip.so "${@/#/scratch/libs/}"; do
    GPR_LIB_PATH=$PWD/$lib timeout -k 10 300 python scratch/ab_sub.py "$W" "$S" 2>&1 | tail -1
  done
done
