#!/bin/bash
# interleaved A/B of the in-tree library against scratch/libs/$1 on bench.py's timed region: ab2.sh other.so "<workloads>" "<S values>"
O="$1"; W="${2:-m4 m8 m4c2 m8s}"; S="${3:-1 4}"
for rep in 1 2; do
  for lib in gymnasium-planar-robotics_amd/libgpr_hip.so scratch/libs/$O; do
    GPR_LIB_PATH=$PWD/$lib timeout -k 10 300 python scratch/ab_sub.py "$W" "$S" 2>&1 | tail -1
  done
done
