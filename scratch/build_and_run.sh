#!/bin/bash
# usage: scratch/build_and_run.sh '<gpu command>' [timeout]
set -e
cd /root/repo
if make -C gymnasium-planar-robotics_amd/csrc -j8 2>&1 | grep -E "error|warning:"; then echo BUILD PROBLEM; exit 1; fi
make -C oracle 2>&1 | grep -E "error|warning" && exit 1
/usr/local/graft/bin/gpurun --timeout ${2:-1100} -- "$1"
