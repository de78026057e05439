#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_mfma.py tests/test_segnet.py -x -q -m gpu -k "token or codes or labels_from" 2>&1 | tail -2
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tk
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -- python3 $ROOT/tools/profile_ppnet.py 256 > /dev/null 2>&1
grep "tokenizer_fused\|gennet_dec_final\|gennet_trunk" $(find /tmp/tk -name "*kernel_stats.csv" | head -1) | cut -c1-140
