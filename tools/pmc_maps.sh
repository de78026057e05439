#!/bin/bash
# SQ counter passes for the stage-B kernels (diagnostic): python tools/split_timing.py under rocprofv3 --pmc, one pass per group
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_IFETCH SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $grp -d /tmp/pm --output-format csv -- python3 /root/repo/tools/split_timing.py > /dev/null 2>&1
  python3 /root/repo/tools/pmc_avg.py /tmp/pm | grep "edage_maps"
done
