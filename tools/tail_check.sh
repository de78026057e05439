#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_plan_tail.py tests/test_ppnet_config3.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/tail.log 2>&1 || { tail -40 gpurun_out/tail.log; exit 1; }
tail -2 gpurun_out/tail.log
timeout -k 10 300 python bench.py --steps 6 --warmup 2 > gpurun_out/tail_bench.json 2> gpurun_out/tail_bench.err || { tail -20 gpurun_out/tail_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/tail_bench.json").read().strip().splitlines()[-1])
p = d["ppnet"]
print("instances/s", d["value"], "plans/s", p["value"], {k: v for k, v in p.items() if k.startswith("ms_") or "rate" in k or "success" in k})
PY
