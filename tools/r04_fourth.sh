#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_segnet.py tests/test_ppnet_config3.py tests/test_gpu_bench.py -x -q -m gpu > $OUT/t_seg.log 2>&1 || { tail -40 $OUT/t_seg.log; exit 1; }
tail -3 $OUT/t_seg.log
timeout -k 10 600 python bench.py > $OUT/bench_mid.json 2> $OUT/bench_mid.err || { tail -30 $OUT/bench_mid.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_mid.json").read().strip().splitlines()[-1])
p = d["ppnet"]
print("instances/s", d["value"], "roofline", d["roofline"]["frac"], "group", d["path_group"], "plans/s", p["value"], "ms/batch", p["ms_per_batch"])
print("parity", json.dumps(p.get("parity"))[:600])
print("e2e", json.dumps({k: v for k, v in d.get("end_to_end_r512", {}).items() if k not in ("config",)})[:3500])
PY
