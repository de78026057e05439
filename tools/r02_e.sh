#!/bin/bash
# one GPU call: the new kernels' tests, the SegNet parity tests, a bench line
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_mfma.py -x -q -m gpu -k "nat128 or tokenizer or heatmap" > gpurun_out/e_tests.log 2>&1 || { tail -30 gpurun_out/e_tests.log; exit 1; }
tail -3 gpurun_out/e_tests.log
timeout -k 10 500 python -m pytest tests/test_segnet.py tests/test_ppnet_config3.py -x -q -m gpu > gpurun_out/e_seg.log 2>&1 || { tail -30 gpurun_out/e_seg.log; exit 1; }
tail -2 gpurun_out/e_seg.log
timeout -k 10 300 python bench.py --steps 6 --warmup 2 > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err || { tail -20 gpurun_out/e_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/e_bench.json").read().strip().splitlines()[-1])
p = d["ppnet"]
print("instances/s", d["value"], "plans/s", p["value"], "ms/batch", p["ms_per_batch"], {k: v for k, v in p.items() if k.startswith("ms_")})
PY
