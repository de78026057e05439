#!/bin/bash
# per-kernel time of one steady-state PPNet batch (kernel trace), default path
OUT=$PWD/gpurun_out/r03; mkdir -p $OUT; ROOT=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $ROOT/tools/profile_ppnet.py 256 > $OUT/pp.log 2>&1
python3 $ROOT/tools/kernel_breakdown.py /tmp/pp extract_paths_kernel 2 45 > $OUT/ppnet_kernel_breakdown_b256_${1:-x}.txt
grep "ms per batch" $OUT/pp.log >> $OUT/ppnet_kernel_breakdown_b256_${1:-x}.txt
cat $OUT/ppnet_kernel_breakdown_b256_${1:-x}.txt | cut -c1-150 | head -40
