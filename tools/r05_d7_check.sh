#!/bin/bash
# dense-group attention kernel after the bookkeeping rewrite: parity, then per-shape timings
mkdir -p gpurun_out/r05
timeout -k 10 400 python -m pytest tests/test_gpu_na.py -x -q -m gpu > gpurun_out/r05/d7_tests.log 2>&1 || { tail -30 gpurun_out/r05/d7_tests.log; exit 1; }
tail -1 gpurun_out/r05/d7_tests.log
timeout -k 10 300 python tools/na_timing.py 2>&1 | grep side | tee gpurun_out/r05/na2d_timing_d7.txt
