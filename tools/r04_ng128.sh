#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_natgemm.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do timeout -k 10 200 python tools/natgemm_timing.py 2>&1 | grep "^level\|^DiNAT" | grep "proj\|fc2\|DiNAT" | sed -e 's/HBM floor.*//'; done > gpurun_out/r04/ng128_prefetch.txt 2>&1
cut -c1-150 gpurun_out/r04/ng128_prefetch.txt
