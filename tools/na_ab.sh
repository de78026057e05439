#!/bin/bash
# NA kernels: parity tests under both routings, then per-shape timings
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_na.py -x -q -m gpu > gpurun_out/na_ab.log 2>&1 || { tail -30 gpurun_out/na_ab.log; exit 1; }
tail -2 gpurun_out/na_ab.log
PPNET_NA_MFMA=1 timeout -k 10 400 python -m pytest tests/test_gpu_na.py -x -q -m gpu > gpurun_out/na_ab_all.log 2>&1 || { tail -30 gpurun_out/na_ab_all.log; exit 1; }
tail -2 gpurun_out/na_ab_all.log
echo "== default routing"; timeout -k 10 200 python tools/na_timing.py 2>&1 | grep side
echo "== PPNET_NA_MFMA=1 (every shape on the MFMA kernel)"; PPNET_NA_MFMA=1 timeout -k 10 200 python tools/na_timing.py 2>&1 | grep side
echo "== PPNET_NA_VALU=1 (every shape on the VALU kernel)"; PPNET_NA_VALU=1 timeout -k 10 200 python tools/na_timing.py 2>&1 | grep side
