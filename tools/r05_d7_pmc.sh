#!/bin/bash
# dense-group attention kernel at 64 x 64, dilation 16 (NA_SHAPE): traffic and issue counters, separate PMC passes
ROOT=$PWD; mkdir -p gpurun_out/r05; export TMPDIR=/tmp; cd /tmp
export NA_SHAPE=${NA_SHAPE:-64,16}
pass() {
  local name=$1; shift
  rm -rf /tmp/d7_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/d7_$name -- python3 $ROOT/tools/na_timing.py > $ROOT/gpurun_out/r05/d7_$name.log 2>&1 || { tail -5 $ROOT/gpurun_out/r05/d7_$name.log; exit 1; }
  python3 $ROOT/tools/pmc_avg.py /tmp/d7_$name | grep -i "dense7" >> $ROOT/gpurun_out/r05/d7_pmc_${NA_SHAPE/,/_}.txt
  echo "pass $name done"
}
rm -f $ROOT/gpurun_out/r05/d7_pmc_${NA_SHAPE/,/_}.txt
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
pass b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU
pass c FETCH_SIZE
pass d WRITE_SIZE
pass e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass f TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
cat $ROOT/gpurun_out/r05/d7_pmc_${NA_SHAPE/,/_}.txt | cut -c1-600
