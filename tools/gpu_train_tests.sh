#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/train.log 2>&1 || { tail -40 gpurun_out/train.log; exit 1; }
tail -3 gpurun_out/train.log
