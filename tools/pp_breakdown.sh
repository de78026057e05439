#!/bin/bash
# per-kernel time of one steady-state PPNet batch (run on the GPU box from the repo root)
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp && rm -rf /tmp/pp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $ROOT/tools/profile_ppnet.py 256 > $ROOT/gpurun_out/pp.log 2>&1 || { tail -20 $ROOT/gpurun_out/pp.log; exit 1; }
python3 $ROOT/tools/kernel_breakdown.py /tmp/pp extract_paths_kernel 2 45 > $ROOT/gpurun_out/pp_breakdown.txt
grep "ms per batch" $ROOT/gpurun_out/pp.log >> $ROOT/gpurun_out/pp_breakdown.txt
python3 $ROOT/tools/kernel_sequence.py /tmp/pp extract_paths_kernel 2 > $ROOT/gpurun_out/pp_sequence.txt
