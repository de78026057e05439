#!/bin/bash
# halo attention kernel: 2 x 8 query blocks (8 key tiles) against 4 x 4 (10): parity under both, then per-shape timings alternating
mkdir -p gpurun_out/r05
timeout -k 10 400 python -m pytest tests/test_gpu_na.py -x -q -m gpu > gpurun_out/r05/halo28_tests.log 2>&1 || { tail -30 gpurun_out/r05/halo28_tests.log; exit 1; }
tail -1 gpurun_out/r05/halo28_tests.log
PPNET_NA_HALO_BLOCK=4x4 timeout -k 10 400 python -m pytest tests/test_gpu_na.py -x -q -m gpu > gpurun_out/r05/halo44_tests.log 2>&1 || { tail -30 gpurun_out/r05/halo44_tests.log; exit 1; }
tail -1 gpurun_out/r05/halo44_tests.log
for rep in 1 2; do
for sh in 64,1 32,1 16,1; do
  echo "== $sh 2x8"; NA_SHAPE=$sh timeout -k 10 100 python tools/na_timing.py 2>&1 | grep side
  echo "== $sh 4x4"; NA_SHAPE=$sh PPNET_NA_HALO_BLOCK=4x4 timeout -k 10 100 python tools/na_timing.py 2>&1 | grep side
done
done | tee gpurun_out/r05/halo_block_timing.txt
