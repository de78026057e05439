#!/bin/bash
# HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of a PPNet batch: the "traffic against algorithmic bytes" check
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r04; cd /tmp
OUT=$ROOT/gpurun_out/r04/ppnet_traffic.txt; : > $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ppt
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/ppt -- python3 $ROOT/tools/profile_ppnet.py 256 > /dev/null 2>&1 || { echo "pass failed: $c" >> $OUT; break; }
  echo "== $c (KiB per launch, averaged over the launches of the run)" >> $OUT
  python3 $ROOT/tools/pmc_avg.py /tmp/ppt 2>&1 | grep -v "at::native\|rocclr" >> $OUT
done
cut -c1-170 $OUT
