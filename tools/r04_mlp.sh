#!/bin/bash
# fused-MLP iteration: parity test, then the timing table (fused vs library vs two-kernel form)
OUT=gpurun_out/r04; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_natgemm.py -x -q -m gpu -k "fused_mlp" > $OUT/t_mlp.log 2>&1 || { tail -40 $OUT/t_mlp.log; exit 1; }
tail -2 $OUT/t_mlp.log
timeout -k 10 300 python tools/natgemm_timing.py > $OUT/natgemm_timing_mlp.txt 2>&1 || { tail -20 $OUT/natgemm_timing_mlp.txt; exit 1; }
grep -i "fused\|dense half" $OUT/natgemm_timing_mlp.txt
