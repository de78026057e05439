#!/bin/bash
# the config-5 step (512 x 512) under the GEMM gate: default (vendor GEMM from C = 512), own kernels everywhere, vendor on level 3 only
mkdir -p gpurun_out/r04
{ for rep in 1 2; do
echo -n "default                                   "; timeout -k 10 200 python tools/profile_e2e.py 2>&1 | tail -1
echo -n "PPNET_LIBRARY_GEMM_FROM_C=1073741824 (own) "; PPNET_LIBRARY_GEMM_FROM_C=1073741824 timeout -k 10 200 python tools/profile_e2e.py 2>&1 | tail -1
echo -n "PPNET_LIBRARY_GEMM_FROM_C=1024             "; PPNET_LIBRARY_GEMM_FROM_C=1024 timeout -k 10 200 python tools/profile_e2e.py 2>&1 | tail -1
done; } > gpurun_out/r04/e2e_gate.txt 2>&1
cat gpurun_out/r04/e2e_gate.txt
