for rep in 1 2 3; do
  for v in base mw7 mw8; do
    if [ $v = base ]; then unset PPNET_HIP_LIB; else export PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_$v.so; fi
    python bench.py --no-ppnet --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep $rep  %.2f M instances/s  %.4f ms/step  kernel %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
python tools/maps_alone.py 2>&1 | tail -3
PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_mw7.so python tools/maps_alone.py 2>&1 | tail -3
PPNET_HIP_LIB=$PWD/ppnet_amd/libppnet_hip_mw8.so python tools/maps_alone.py 2>&1 | tail -3
