#!/bin/bash
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r04; cd /tmp; rm -rf /tmp/pe
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -- python3 $ROOT/tools/profile_e2e.py > $ROOT/gpurun_out/r04/pe.log 2>&1 || { tail -5 $ROOT/gpurun_out/r04/pe.log; exit 1; }
python3 $ROOT/tools/kernel_breakdown.py /tmp/pe extract_paths_kernel 0 45 > $ROOT/gpurun_out/r04/e2e_breakdown.txt
grep "ms per step" $ROOT/gpurun_out/r04/pe.log >> $ROOT/gpurun_out/r04/e2e_breakdown.txt
cut -c1-150 $ROOT/gpurun_out/r04/e2e_breakdown.txt | head -48
