#!/bin/bash
# The A/B knobs select older / library paths of the same ops: every one of them must still pass the tests of the op it touches.
# Run on the GPU box from the repo root; one pytest process per knob.  PART=1 | 2 runs half of the list (a gpurun call holds 20 minutes).
mkdir -p gpurun_out
rc=0
n=0
run() { # knob, test selection...
  local knob=$1; shift
  n=$((n + 1))
  if [ -n "$PART" ] && [ $(( (n + 1) % 2 + 1 )) != "$PART" ]; then return; fi
  if env $knob timeout -k 10 600 python -m pytest "$@" -x -q -m gpu > gpurun_out/knob.log 2>&1; then echo "ok    $knob   $(tail -1 gpurun_out/knob.log)"; else echo "FAIL  $knob"; tail -15 gpurun_out/knob.log; rc=1; fi
}
G="tests/test_gennet_golden.py tests/test_ppnet_config3.py"
S="tests/test_segnet.py tests/test_ppnet_config3.py"
run PPNET_TRUNK_512=1 $G tests/test_gpu_mfma.py
run PPNET_TRUNK_VALU=1 $G tests/test_gpu_mfma.py
run PPNET_TO1_VALU=1 $G tests/test_segnet.py
run PPNET_GENNET_UNFUSED_TAIL=1 $G
run PPNET_GENNET_UNFUSED=1 $G
run PPNET_LIBRARY_TRUNK=1 $G
run PPNET_NA_VALU=1 tests/test_gpu_na.py $S
run PPNET_NA_NO_DENSE7=1 tests/test_gpu_na.py $S
run PPNET_LIBRARY_NAT128=1 $S
run PPNET_TOKENIZER_TWO_KERNELS=1 $S
run PPNET_LIBRARY_TOKENIZER=1 $S
run PPNET_NO_FOLD=1 $S
run PPNET_NA_HALO16=0 tests/test_gpu_na.py $S
run PPNET_NO_LN_FOLD=1 $S tests/test_gpu_natgemm.py
run PPNET_LIBRARY_GEMM=1 $S
run PPNET_NAT_GEMM128=0 $S tests/test_gpu_natgemm.py
run PPNET_NAT_GEMM128=all $S tests/test_gpu_natgemm.py
run PPNET_NO_SMALL_GEMM=1 $S tests/test_gpu_mfma.py
run PPNET_NO_FUSED_MLP=1 $S tests/test_gpu_natgemm.py
run PPNET_LIBRARY_GEMM_FROM_C=1073741824 $S
run PPNET_LIBRARY_GEMM_FROM_C=1024 $S
run PPNET_UPER_UNFUSED_RESIZE=1 tests/test_segnet.py
run PPNET_NA_HALO_BLOCK=4x4 tests/test_gpu_na.py $S
run PPNET_NAT_LN=old $S tests/test_gpu_natgemm.py
run PPNET_NAT_ACC=old $S tests/test_gpu_natgemm.py
exit $rc
