#!/bin/bash
# nat128_ln_mlp_kernel with one part removed per build (make -C ppnet_amd/csrc c128abl): where its time goes
mkdir -p gpurun_out/r04
{
for v in base 1 2 3 4 8 16 24 32; do
  if [ $v = base ]; then unset PPNET_HIP_LIB; else export PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_c128abl$v.so; fi
  echo -n "C128_ABL=$v  "; timeout -k 10 120 python tools/nat128_timing.py 2>&1 | grep "ln+mlp"
done
} > gpurun_out/r04/c128_abl.txt 2>&1
cat gpurun_out/r04/c128_abl.txt
