ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/na_f /tmp/na_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/na_f -- python3 $ROOT/tools/na_timing.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_f | grep na2d
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/na_w -- python3 $ROOT/tools/na_timing.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_w | grep na2d
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/na_h -- python3 $ROOT/tools/na_timing.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_h | grep na2d
