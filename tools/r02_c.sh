#!/bin/bash
OUT=$PWD/gpurun_out/r02c; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_mfma.py tests/test_gennet_golden.py tests/test_ppnet_config3.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest exit $?"
tail -12 $OUT/pytest.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "
import json
d=json.load(open('$OUT/bench.json')); p=d['ppnet']
print({k:p[k] for k in ('value','ms_per_batch','ms_segnet','ms_gennet','ms_tail')})"
PPNET_LIBRARY_TRUNK=1 python bench.py --no-cpu-baseline --steps 3 > $OUT/bench_libtrunk.json 2> $OUT/bench_libtrunk.err
python -c "
import json
d=json.load(open('$OUT/bench_libtrunk.json')); p=d['ppnet']
print('library trunk:', {k:p[k] for k in ('value','ms_per_batch','ms_segnet','ms_gennet','ms_tail')})"
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/profile_ppnet.py 256 > $OUT/pp.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_breakdown.py /tmp/pp extract_paths_kernel 2 40 > $OUT/ppnet_kernel_breakdown.txt; cat $OUT/ppnet_kernel_breakdown.txt
