#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu -k "nat128" 2>&1 | tail -3
for rep in 1 2; do timeout -k 10 120 python tools/nat128_timing.py 2>&1 | grep "fused"; timeout -k 10 120 python tools/nat128_proj_timing.py 2>&1 | tail -4; done > gpurun_out/r04/c128_timing.txt 2>&1
cat gpurun_out/r04/c128_timing.txt
