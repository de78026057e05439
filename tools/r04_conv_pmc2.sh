#!/bin/bash
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r04; cd /tmp
OUT=$ROOT/gpurun_out/r04/conv_pmc_cbmajor.txt; : > $OUT
timeout -k 10 300 $ROOT/tools/micro/gemm_bench >> $OUT 2>&1 || { tail -30 $OUT; exit 1; }
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/cpmc
  timeout -k 10 90 rocprofv3 --pmc $c --output-format csv -d /tmp/cpmc -- $ROOT/tools/micro/gemm_bench conv64 > /dev/null 2>&1 || { echo "counter pass failed: $c" >> $OUT; break; }
  echo "== $c" >> $OUT
  python3 $ROOT/tools/pmc_avg.py /tmp/cpmc 2>&1 | grep "gemm_bf16" >> $OUT
done
tail -22 $OUT
