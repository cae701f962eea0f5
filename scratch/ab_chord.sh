#!/bin/bash
# A/B of the chord pre-test build against -DGPR_NO_CHORD (scratch/libs/nochord.so) on bench.py's timed region, interleaved, one GPU call
W="${1:-m4 m8 m16 m4s m16s}"; S="${2:-1 4}"
for rep in 1 2; do
  for lib in gymnasium-planar-robotics_amd/libgpr_hip.so scratch/libs/nochord.so; do
    GPR_LIB_PATH=$PWD/$lib timeout -k 10 300 python scratch/ab_sub.py "$W" "$S" 2>&1 | tail -1
  done
done
