#!/bin/bash
# Generator throughput against the length of the timed region (one box, back to back): the chip leaves its idle clocks only after
# tens of milliseconds of load and meets its power limit after seconds.  warmup = steps / 4 (at least 3).
for st in 20 100 400 2000 4000 16000; do
  w=$(( st / 4 )); [ $w -lt 3 ] && w=3
  for rep in 1 2; do
    python bench.py --no-ppnet --no-cpu-baseline --steps $st --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps=%6d warmup=%5d  %.2f M instances/s  %.4f ms/step  maps kernel %.4f ms  roofline %.3f' % ($st, $w, d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
