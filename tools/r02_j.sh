#!/bin/bash
OUT=$PWD/gpurun_out/r02j; mkdir -p $OUT
export TMPDIR=/tmp
run() { # name env...
  name=$1; shift
  env "$@" python bench.py --no-ppnet --no-cpu-baseline --steps 60 --warmup 10 > $OUT/$name.json 2>$OUT/$name.err
  python -c "
import json
d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M/s  step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])" || tail -3 $OUT/$name.err
}
for i in 1 2; do
run g1_w2_$i BENCH_PATH_GROUP=1
run g10_w2_$i BENCH_PATH_GROUP=10
run g1_w4_$i BENCH_PATH_GROUP=1 PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_w4.so
run g10_w4_$i BENCH_PATH_GROUP=10 PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_w4.so
run g5_w4_$i BENCH_PATH_GROUP=5 PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_w4.so
done
