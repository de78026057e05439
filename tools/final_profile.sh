#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): bench line, kernel stats, HBM traffic, SQ counters,
# PPNet per-kernel breakdown.  Outputs land in gpurun_out/final/; copy the summaries to profiles/.
set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 50 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $ROOT/bench.py --no-cpu-baseline --no-ppnet --steps 50 > $OUT/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ppnet > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ppnet > $OUT/pw.log 2>&1
cd $ROOT && python tools/collect_traffic.py /tmp/pmc_fetch /tmp/pmc_write r01 > $OUT/traffic.log && cp profiles/traffic_maps_kernel.json $OUT/ && cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/sq -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ppnet > $OUT/sq.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/sq > $OUT/pmc_sq_counters.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $ROOT/tools/profile_ppnet.py 256 > $OUT/pp.log 2>&1
python3 $ROOT/tools/kernel_breakdown.py /tmp/pp extract_paths_kernel 2 30 > $OUT/ppnet_kernel_breakdown_b256.txt
grep "ms per batch" $OUT/pp.log >> $OUT/ppnet_kernel_breakdown_b256.txt
tail -c 600 $OUT/bench.json
