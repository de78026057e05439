#!/bin/bash
# SQ counters of the NA kernels (separate PMC passes)
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/na_a -- python3 $ROOT/tools/na_timing.py > $ROOT/gpurun_out/h_a.log 2>&1 || { tail -5 $ROOT/gpurun_out/h_a.log; exit 1; }
python3 $ROOT/tools/pmc_avg.py /tmp/na_a | grep -i "na2d" > $ROOT/gpurun_out/h_pmc_a.txt
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d /tmp/na_b -- python3 $ROOT/tools/na_timing.py > $ROOT/gpurun_out/h_b.log 2>&1 || { tail -5 $ROOT/gpurun_out/h_b.log; exit 1; }
python3 $ROOT/tools/pmc_avg.py /tmp/na_b | grep -i "na2d" > $ROOT/gpurun_out/h_pmc_b.txt
cat $ROOT/gpurun_out/h_pmc_a.txt $ROOT/gpurun_out/h_pmc_b.txt | cut -c1-400
