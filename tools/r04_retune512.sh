#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python tools/retune_gemms.py fresh gpurun_out/r04/tuned_r512.csv 512 > gpurun_out/r04/retune512.log 2>&1; echo "rc $?"; tail -5 gpurun_out/r04/retune512.log
timeout -k 10 200 python tools/retune_gemms.py 512 >> gpurun_out/r04/retune512.log 2>&1; tail -1 gpurun_out/r04/retune512.log
PPNET_TUNED_TABLE=$PWD/gpurun_out/r04/tuned_r512.csv timeout -k 10 200 python tools/retune_gemms.py 512 >> gpurun_out/r04/retune512.log 2>&1; tail -1 gpurun_out/r04/retune512.log
