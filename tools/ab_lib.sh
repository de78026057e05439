#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: usage tools/ab_lib.sh path/to/other.so [rounds]
OTHER=$1; N=${2:-3}
for i in $(seq $N); do
  for lib in "" "$OTHER"; do
    if [ -n "$lib" ]; then export PPNET_HIP_LIB=$PWD/$lib; else unset PPNET_HIP_LIB; fi
    echo -n "${lib:-default}: "
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ppnet | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
