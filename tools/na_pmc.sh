#!/bin/bash
# SQ counters of the NA kernels (separate PMC passes).  NA_SHAPE="side,dilation" restricts tools/na_timing.py to one shape.
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/na_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/na_$name -- python3 $ROOT/tools/na_timing.py > $ROOT/gpurun_out/h_$name.log 2>&1 || { tail -5 $ROOT/gpurun_out/h_$name.log; exit 1; }
  python3 $ROOT/tools/pmc_avg.py /tmp/na_$name | grep -i "na2d" > $ROOT/gpurun_out/h_pmc_$name.txt
}
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
pass b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU
pass c SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
cat $ROOT/gpurun_out/h_pmc_a.txt $ROOT/gpurun_out/h_pmc_b.txt $ROOT/gpurun_out/h_pmc_c.txt | cut -c1-500
