#!/bin/bash
# what the matrix pipe waits on in the SETR-UP head's last convolution (gemm_bf16_kernel<1, 4> at 64 x 64, batch 256):
# LDS instruction / bank-conflict / busy counters, vector-memory instruction count, issue-stall buckets — one counter group per pass
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r05; cd /tmp
OUT=$ROOT/gpurun_out/r05/conv_head_pmc.txt; : > $OUT
timeout -k 10 120 $ROOT/tools/micro/gemm_bench conv64 >> $OUT 2>&1 || { cat $OUT; exit 1; }
for c in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_WAVES"; do
  rm -rf /tmp/cpmc
  timeout -k 10 90 rocprofv3 --pmc $c --output-format csv -d /tmp/cpmc -- $ROOT/tools/micro/gemm_bench conv64 > /dev/null 2>&1 || { echo "counter pass failed: $c" >> $OUT; continue; }
  echo "== $c" >> $OUT
  python3 $ROOT/tools/pmc_avg.py /tmp/cpmc 2>&1 | grep "gemm_bf16" >> $OUT
done
cat $OUT
