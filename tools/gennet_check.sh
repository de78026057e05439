#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mfma.py tests/test_gennet_golden.py tests/test_ppnet_config3.py -x -q -m gpu > gpurun_out/gennet.log 2>&1 || { tail -40 gpurun_out/gennet.log; exit 1; }
tail -2 gpurun_out/gennet.log
bash tools/ab_bench.sh PPNET_GENNET_UNFUSED
