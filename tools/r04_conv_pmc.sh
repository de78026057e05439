#!/bin/bash
# L2 / HBM traffic of the head's last convolution (gemm_bf16_kernel<1, 4> at 64 x 64) against a dense GEMM of the same core
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r04; cd /tmp
OUT=$ROOT/gpurun_out/r04/conv_pmc.txt; : > $OUT
timeout -k 10 120 $ROOT/tools/micro/gemm_bench conv64 >> $OUT 2>&1 || { cat $OUT; exit 1; }
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/cpmc
  timeout -k 10 90 rocprofv3 --pmc $c --output-format csv -d /tmp/cpmc -- $ROOT/tools/micro/gemm_bench conv64 > /dev/null 2>&1 || { echo "counter pass failed: $c" >> $OUT; break; }
  echo "== $c" >> $OUT
  python3 $ROOT/tools/pmc_avg.py /tmp/cpmc 2>&1 | grep "gemm_bf16" >> $OUT
done
cat $OUT
