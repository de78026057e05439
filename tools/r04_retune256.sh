#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python tools/retune_gemms.py fresh gpurun_out/r04/tuned_r256.csv 256 > gpurun_out/r04/retune256.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r04/retune256.log
for i in 1 2; do
timeout -k 10 200 python tools/retune_gemms.py 256 2>&1 | tail -1
PPNET_TUNED_TABLE=$PWD/gpurun_out/r04/tuned_r256.csv timeout -k 10 200 python tools/retune_gemms.py 256 2>&1 | tail -1
done
timeout -k 10 200 python tools/retune_gemms.py 512 2>&1 | tail -1
timeout -k 10 200 python tools/profile_e2e.py 2>&1 | tail -1
