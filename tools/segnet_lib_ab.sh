#!/bin/bash
# SegNet batch time under two builds of the library, alternating PROCESSES on one box: tools/segnet_lib_ab.sh path/to/other.so [rounds]
OTHER=$1; N=${2:-3}
for i in $(seq $N); do
  for lib in "" "$OTHER"; do
    if [ -n "$lib" ]; then export PPNET_HIP_LIB=$PWD/$lib; else unset PPNET_HIP_LIB; fi
    echo -n "${lib:-this build}: "
    python tools/segnet_env_ab.py 2>&1 | grep "round 3" | sed 's/round 3 default *//'
  done
done
