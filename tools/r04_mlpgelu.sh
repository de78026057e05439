#!/bin/bash
# the fused MLP kernel (C = 256) under its GELU forms (make -C ppnet_amd/csrc mlpgelu): base = logistic fit, two values per instruction
mkdir -p gpurun_out/r04
{ for rep in 1 2; do for v in base mlpgelu1 mlpgelu2; do
  if [ $v = base ]; then unset PPNET_HIP_LIB; else export PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_$v.so; fi
  timeout -k 10 120 python tools/mlp_timing.py 2>&1 | tail -1
done; done; } > gpurun_out/r04/mlpgelu_ab.txt 2>&1
unset PPNET_HIP_LIB
cat gpurun_out/r04/mlpgelu_ab.txt
PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_mlpgelu2.so timeout -k 10 200 python -m pytest tests/test_gpu_natgemm.py -x -q -m gpu -k fused_mlp 2>&1 | tail -2
