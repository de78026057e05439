#!/bin/bash
# A/B of several builds of the library on ONE box, alternating: usage tools/ab_libs.sh ROUNDS lib1.so lib2.so ... ("default" = the shipped one)
N=$1; shift
for i in $(seq $N); do
  for lib in "$@"; do
    if [ "$lib" != "default" ]; then export PPNET_HIP_LIB=$PWD/$lib; else unset PPNET_HIP_LIB; fi
    echo -n "$lib: "
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ppnet | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
