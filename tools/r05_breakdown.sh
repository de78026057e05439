#!/bin/bash
# PPNet: per-kernel time of one steady-state batch under the kernel tracer
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r05; cd /tmp; rm -rf /tmp/pp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $ROOT/tools/profile_ppnet.py 256 > $ROOT/gpurun_out/r05/pp.log 2>&1 || { tail -5 $ROOT/gpurun_out/r05/pp.log; exit 1; }
python3 $ROOT/tools/kernel_breakdown.py /tmp/pp extract_paths_kernel 2 45 > $ROOT/gpurun_out/r05/breakdown.txt
grep "ms per batch" $ROOT/gpurun_out/r05/pp.log >> $ROOT/gpurun_out/r05/breakdown.txt
cut -c1-150 $ROOT/gpurun_out/r05/breakdown.txt | head -40
