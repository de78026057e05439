#!/bin/bash
mkdir -p gpurun_out/r04
{ for s in 1 2 3 4; do
  PPNET_STREAMS=$s timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppnet-fp32 --no-end-to-end --segnet dinat_setr --ppnet-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppnet']; print('PPNET_STREAMS=$s  plans/s %.0f  ms/batch %.2f' % (p['value'], p['ms_per_batch']))"
done; } > gpurun_out/r04/streams.txt 2>&1
cat gpurun_out/r04/streams.txt
