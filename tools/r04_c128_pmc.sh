#!/bin/bash
# LDS bank conflicts and instruction counts of level 0's three kernels: the round's build against the one before its operand remap
# (ppnet_amd/libppnet_hip_c128old.so, built from commit 5accc1e's nat_c128.hip)
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r04; cd /tmp
OUT=$ROOT/gpurun_out/r04/c128_pmc.txt; : > $OUT
for v in new old; do
  if [ $v = old ]; then export PPNET_HIP_LIB=$ROOT/ppnet_amd/libppnet_hip_c128old.so; else unset PPNET_HIP_LIB; fi
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    rm -rf /tmp/c128p
    timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d /tmp/c128p -- python3 $ROOT/tools/nat128_timing.py > /dev/null 2>&1 || { echo "pass failed: $v $c" >> $OUT; break 2; }
    echo "== $v build: $c" >> $OUT
    python3 $ROOT/tools/pmc_avg.py /tmp/c128p 2>&1 | grep "nat128" >> $OUT
  done
done
cut -c1-200 $OUT
