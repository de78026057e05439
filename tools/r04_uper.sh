#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_segnet.py -x -q -m gpu > gpurun_out/r04/uper_tests.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r04/uper_tests.log
timeout -k 10 300 python tools/uper_ab.py > gpurun_out/r04/uper_ab.txt 2>&1; echo "ab rc $?"; cat gpurun_out/r04/uper_ab.txt
