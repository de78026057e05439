#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): bench line (default window and the driver's own command, plain
# and under the kernel tracer), kernel stats, HBM traffic and SQ counters of the maps kernel, PPNet per-kernel breakdown, matrix-pipe
# counters of the MFMA kernels, NA kernel timings and counters, the NAT projection kernels against the vendor GEMM, training steps.
# Outputs land in gpurun_out/final/; copy the summaries to profiles/ (tools/final_profile.sh TAG names them).
# PART=a (bench lines, generator kernel stats / traffic / SQ counters, PPNet breakdown + matrix-pipe counters), PART=b (attention,
# GEMM and MLP tables, A/B, training, graph latency, sweep) or unset (everything) — one gpurun call holds 20 minutes.
set -e
TAG=${1:-r05}
ROOT=$PWD; OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" != "b" ]; then
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# the driver's exact round-end command, for a like-for-like comparison with BENCH_rNN.json — plain and under the kernel tracer
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksd -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppnet > $OUT/ksd.log 2>&1
cp $(find /tmp/ksd -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_driver_cmd.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $ROOT/bench.py --no-cpu-baseline --no-ppnet > $OUT/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ppnet > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ppnet > $OUT/pw.log 2>&1
cd $ROOT && python tools/collect_traffic.py /tmp/pmc_fetch /tmp/pmc_write $TAG > $OUT/traffic.log && cp profiles/traffic_maps_kernel.json $OUT/ && cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/sq -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ppnet > $OUT/sq.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/sq > $OUT/pmc_sq_counters.txt
# PPNet: per-kernel time of one steady-state batch, then the matrix-pipe counters of the same run (separate PMC pass)
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $ROOT/tools/profile_ppnet.py 256 > $OUT/pp.log 2>&1
python3 $ROOT/tools/kernel_breakdown.py /tmp/pp extract_paths_kernel 2 45 > $OUT/ppnet_kernel_breakdown_b256.txt
grep "ms per batch" $OUT/pp.log >> $OUT/ppnet_kernel_breakdown_b256.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/ppm -- python3 $ROOT/tools/profile_ppnet.py 256 > $OUT/ppm.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/ppm | grep -i "gemm\|na2d\|gennet\|nat128\|Cijk" > $OUT/ppnet_pmc_mfma.txt || true
fi
if [ "$PART" != "a" ]; then
cd /tmp
# NA kernels on every (level, dilation) shape + their counters
python3 $ROOT/tools/na_timing.py > $OUT/na_timing.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/na_sq -- python3 $ROOT/tools/na_timing.py > $OUT/na_sq.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_sq | grep na2d > $OUT/na_pmc_sq.txt || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/na_f -- python3 $ROOT/tools/na_timing.py > $OUT/na_f.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_f | grep na2d > $OUT/na_pmc_fetch.txt || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/na_w -- python3 $ROOT/tools/na_timing.py > $OUT/na_w.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_w | grep na2d > $OUT/na_pmc_write.txt || true
cd $ROOT
GEMM_ONE_TILE_PER_BLOCK=1 tools/micro/gemm_bench > $OUT/gemm_bench.txt 2>&1 || true
tools/micro/gemm_bench >> $OUT/gemm_bench.txt 2>&1 || true
python3 tools/gemm_vs_lib.py > $OUT/gemm_vs_lib.txt 2>&1 || true
python3 tools/natgemm_timing.py > $OUT/natgemm_timing.txt 2>&1 || true
python3 tools/mlp_timing.py 512 > $OUT/mlp_timing.txt 2>&1 || true
for h in 128 256 1024; do python3 tools/mlp_timing.py $h >> $OUT/mlp_timing.txt 2>&1 || true; done
python3 tools/ppnet_ab.py > $OUT/ppnet_ab.txt 2>&1 || true
NA_SHAPE=64,1 bash tools/na_pmc.sh > $OUT/na2d_halo16_pmc.txt 2>&1 || true
NA_SHAPE=64,1 bash tools/na_fetch.sh >> $OUT/na2d_halo16_pmc.txt 2>&1 || true
python3 tools/train_timing.py > $OUT/train_timing.txt 2>&1 || true
python3 tools/na_bwd_timing.py >> $OUT/train_timing.txt 2>&1 || true
python3 tools/graph_latency.py > $OUT/graph_latency.txt 2>&1 || true
bash tools/steps_sweep.sh > $OUT/steps_sweep.txt 2>&1 || true
fi
[ -f $OUT/bench.json ] && tail -c 900 $OUT/bench.json || true
