#!/bin/bash
OUT=$PWD/gpurun_out/r02d; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_mfma.py tests/test_ppnet_config3.py -m gpu -x -q -s -k "trunk or heatmap" > $OUT/pytest.log 2>&1; echo "pytest exit $?"
tail -8 $OUT/pytest.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "
import json
d=json.load(open('$OUT/bench.json')); p=d['ppnet']
print({k:p[k] for k in ('value','ms_per_batch','ms_segnet','ms_gennet','ms_tail')})"
