#!/bin/bash
# second GPU pass of round 2: MFMA kernel tests, the config-3 parity tests on the MFMA head, bench
OUT=$PWD/gpurun_out/r02b; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_mfma.py tests/test_ppnet_config3.py tests/test_segnet.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest exit $?"
tail -8 $OUT/pytest.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "
import json
d=json.load(open('$OUT/bench.json')); p=d['ppnet']
print({k:p[k] for k in ('value','ms_per_batch','ms_segnet','ms_gennet','ms_tail')})"
PPNET_LIBRARY_CONV=1 python bench.py --no-cpu-baseline > $OUT/bench_libconv.json 2> $OUT/bench_libconv.err
python -c "
import json
d=json.load(open('$OUT/bench_libconv.json')); p=d['ppnet']
print('library conv:', {k:p[k] for k in ('value','ms_per_batch','ms_segnet','ms_gennet','ms_tail')})"
