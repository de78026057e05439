#!/bin/bash
OUT=$PWD/gpurun_out/r02g2; mkdir -p $OUT
for m in 0 1 2; do echo "dbg $m"; PPNET_NA_DBG=$m timeout -k 10 300 python tools/na_timing.py 2>&1 | grep "d  1 pad\|d  2 pad  16"; done
