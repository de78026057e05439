#!/bin/bash
OUT=$PWD/gpurun_out/r02h; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_segnet.py tests/test_ppnet_config3.py tests/test_gennet_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "
import json
d=json.load(open('$OUT/bench.json')); p=d['ppnet']
print({k:p[k] for k in ('value','ms_per_batch','ms_segnet','ms_gennet','ms_tail')})"
