#!/bin/bash
# which part of the fused MLP kernel is the time?  one diagnostic build per removed part (make -C ppnet_amd/csrc abl)
OUT=gpurun_out/r04; mkdir -p $OUT; : > $OUT/mlp_abl.txt
timeout -k 10 100 python tools/mlp_timing.py >> $OUT/mlp_abl.txt 2>&1 || exit 1
for n in 1 2 4 8 16 32 24; do
  PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_abl$n.so timeout -k 10 100 python tools/mlp_timing.py >> $OUT/mlp_abl.txt 2>&1 || { tail $OUT/mlp_abl.txt; exit 1; }
done
timeout -k 10 100 python tools/mlp_timing.py >> $OUT/mlp_abl.txt 2>&1
grep "ms per launch" $OUT/mlp_abl.txt
