#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 1000 python tools/retune_gemms.py fresh gpurun_out/r04/tuned_fp32.csv 256 fp32 > gpurun_out/r04/retune_fp32.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r04/retune_fp32.log
timeout -k 10 300 python tools/retune_gemms.py 256 fp32 2>&1 | tail -1
PPNET_TUNED_TABLE=$PWD/gpurun_out/r04/tuned_fp32.csv timeout -k 10 300 python tools/retune_gemms.py 256 fp32 2>&1 | tail -1
