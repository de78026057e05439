#!/bin/bash
# round 4, second GPU call: the new tests, the stage-A grouping A/B (marker packets per step), one full bench line
OUT=gpurun_out/r04; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_ppnet_config3.py -x -q -m gpu > $OUT/t_new.log 2>&1 || { tail -40 $OUT/t_new.log; exit 1; }
tail -3 $OUT/t_new.log
for rep in 1 2; do
  for g in 1 4 8; do
    BENCH_PATH_GROUP=$g timeout -k 10 200 python bench.py --warmup 480 --steps 2000 --no-ppnet --no-cpu-baseline > $OUT/group_${g}_$rep.json 2> $OUT/group_${g}_$rep.err || { tail -20 $OUT/group_${g}_$rep.err; exit 1; }
  done
done
for g in 1 5; do
  BENCH_PATH_GROUP=$g timeout -k 10 200 python bench.py --warmup 5 --steps 20 --no-ppnet --no-cpu-baseline > $OUT/group_drv_${g}.json 2> $OUT/group_drv_${g}.err || { tail -20 $OUT/group_drv_${g}.err; exit 1; }
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/group_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "group", d["path_group"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "stageA", d["stage_a"]["kernel_ms"])
PY
timeout -k 10 600 python bench.py > $OUT/bench_mid.json 2> $OUT/bench_mid.err || { tail -30 $OUT/bench_mid.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_mid.json").read().strip().splitlines()[-1])
p = d["ppnet"]
print("instances/s", d["value"], "roofline", d["roofline"]["frac"], "plans/s", p["value"], "ms/batch", p["ms_per_batch"])
print("parity", json.dumps(p.get("parity")))
print("e2e", json.dumps({k: v for k, v in d.get("end_to_end_r512", {}).items() if k not in ("config",)})[:3000])
PY
