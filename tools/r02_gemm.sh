#!/bin/bash
# GEMM core micro-benchmark on the GPU box (binary cross-compiled in the build container): one tile per workgroup, then persistent
OUT=$PWD/gpurun_out/r02g; mkdir -p $OUT
echo "== one tile per workgroup (what the convolution entry points launch)" > $OUT/full.txt
GEMM_ONE_TILE_PER_BLOCK=1 timeout -k 10 600 tools/micro/gemm_bench >> $OUT/full.txt 2>&1; echo "exit $?"
echo "== persistent: one workgroup per CU walks the tiles" >> $OUT/full.txt
timeout -k 10 600 tools/micro/gemm_bench >> $OUT/full.txt 2>&1; echo "exit $?"
grep -c ok $OUT/full.txt; grep -c FAIL $OUT/full.txt
