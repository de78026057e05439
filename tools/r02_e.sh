#!/bin/bash
OUT=$PWD/gpurun_out/r02e; mkdir -p $OUT
export TMPDIR=/tmp
# maps kernel: plain vs non-temporal stores, alternating on one box
for i in 1 2 3; do
  python bench.py --no-ppnet --no-cpu-baseline --steps 60 > $OUT/plain$i.json 2>$OUT/plain$i.err
  PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_nt.so python bench.py --no-ppnet --no-cpu-baseline --steps 60 > $OUT/nt$i.json 2>$OUT/nt$i.err
done
tail -3 $OUT/nt1.err
python - <<PY
import json
for k in ("plain","nt"):
    v=[json.load(open("$OUT/%s%d.json"%(k,i))) for i in (1,2,3)]
    print(k, [round(x["value"]/1e6,2) for x in v], [x["roofline"]["kernel_ms"] for x in v])
PY
