#!/bin/bash
# the round-end checks on one box: the whole GPU suite, smoke(), one bench line
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1 || { tail -40 gpurun_out/full_gpu.log; exit 1; }
tail -2 gpurun_out/full_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
p = d["ppnet"]
print("instances/s", d["value"], "roofline", d["roofline"]["frac"], "plans/s", p["value"], "ms/batch", p["ms_per_batch"], {k: v for k, v in p.items() if k.startswith("ms_")})
PY
