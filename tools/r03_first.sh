#!/bin/bash
# start-of-round baseline on one box: host share, the driver's exact bench command, GEMM shapes vs the library, NA timing
OUT=gpurun_out/r03; mkdir -p $OUT
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; } > $OUT/host.txt 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_start.json 2> $OUT/bench_driver_cmd_start.err || { tail -20 $OUT/bench_driver_cmd_start.err; exit 1; }
timeout -k 10 200 python tools/gemm_vs_lib.py > $OUT/gemm_vs_lib_start.txt 2>&1 || { tail $OUT/gemm_vs_lib_start.txt; exit 1; }
timeout -k 10 200 python tools/na_timing.py > $OUT/na_timing_start.txt 2>&1
cat $OUT/host.txt; cut -c1-900 $OUT/bench_driver_cmd_start.json
