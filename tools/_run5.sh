mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_segnet.py tests/test_gpu_plan_tail.py tests/test_ppnet_config3.py tests/test_gennet_golden.py tests/test_gpu_mfma.py -x -q -m gpu > gpurun_out/t5.log 2>&1 || { tail -40 gpurun_out/t5.log; exit 1; }
tail -2 gpurun_out/t5.log
bash tools/pp_breakdown.sh && head -32 gpurun_out/pp_breakdown.txt | cut -c1-120; tail -1 gpurun_out/pp_breakdown.txt
