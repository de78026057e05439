#!/bin/bash
# GEMM core micro-benchmark on the GPU box (binary cross-compiled in the build container)
OUT=$PWD/gpurun_out/r02g; mkdir -p $OUT
timeout -k 10 300 tools/micro/gemm_bench quick > $OUT/quick.txt 2>&1; echo "quick exit $?"; cat $OUT/quick.txt
timeout -k 10 600 tools/micro/gemm_bench > $OUT/full.txt 2>&1; echo "full exit $?"; cat $OUT/full.txt
