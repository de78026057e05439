#!/bin/bash
OUT=$PWD/gpurun_out/r02i; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python -X faulthandler -m pytest tests/test_segnet.py -m gpu -x -q -s -k "512" > $OUT/t512.log 2>&1; echo "exit $?"; tail -5 $OUT/t512.log
