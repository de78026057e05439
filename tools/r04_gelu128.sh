#!/bin/bash
# nat128_ln_mlp_kernel under the GELU forms (make -C ppnet_amd/csrc gelu128): 0 erf (A&S), 1 logistic fit packed, 3 logistic fit scalar,
# 4 polynomial scalar, base = polynomial packed (ships)
mkdir -p gpurun_out/r04
for rep in 1 2; do
for v in base gelu0 gelu1 gelu3 gelu4; do
  if [ $v = base ]; then unset PPNET_HIP_LIB; else export PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_$v.so; fi
  echo -n "$v rep $rep  "; timeout -k 10 120 python tools/nat128_timing.py 2>&1 | grep "ln+mlp"
done; done > gpurun_out/r04/gelu128_ab.txt 2>&1
unset PPNET_HIP_LIB
cat gpurun_out/r04/gelu128_ab.txt
