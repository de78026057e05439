#!/bin/bash
# A/B on ONE box: bench.py with and without an environment knob.  usage: tools/ab_bench.sh KNOB_NAME
mkdir -p gpurun_out
for v in "" 1; do
  if [ -n "$v" ]; then export $1=1; fi
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -20 gpurun_out/ab_$v.err; exit 1; }
  python - "$1" "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[2]).read().strip().splitlines()[-1])
p = d["ppnet"]
print("%s=%s" % (sys.argv[1], sys.argv[2] or "unset"), "instances/s %.0f" % d["value"], "plans/s", p["value"], {k: v for k, v in p.items() if k.startswith("ms_")})
PY
done
