#!/bin/bash
OUT=$PWD/gpurun_out/r02f; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_na.py tests/test_segnet.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest.log
timeout -k 10 300 python tools/na_timing.py > $OUT/na_timing_mfma.txt 2>&1; cat $OUT/na_timing_mfma.txt
PPNET_NA_VALU=1 timeout -k 10 300 python tools/na_timing.py > $OUT/na_timing_valu.txt 2>&1; cat $OUT/na_timing_valu.txt
