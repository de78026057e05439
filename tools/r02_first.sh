#!/bin/bash
# Round 2, first GPU pass: the GPU test suite, the bench line, and the neighbourhood-attention kernel's counters
# (SQ / FETCH_SIZE / WRITE_SIZE in separate PMC passes, as MI355X_MICROARCH.md prescribes) over tools/na_timing.py.
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02a; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
tail -c 1500 $OUT/bench.json
cd /tmp
python3 $ROOT/tools/na_timing.py > $OUT/na_timing.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/na_sq -- python3 $ROOT/tools/na_timing.py > $OUT/na_sq.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_sq > $OUT/na_pmc_sq.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/na_sq2 -- python3 $ROOT/tools/na_timing.py > $OUT/na_sq2.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_sq2 > $OUT/na_pmc_sq2.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/na_f -- python3 $ROOT/tools/na_timing.py > $OUT/na_f.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_f > $OUT/na_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/na_w -- python3 $ROOT/tools/na_timing.py > $OUT/na_w.log 2>&1
python3 $ROOT/tools/pmc_avg.py /tmp/na_w > $OUT/na_pmc_write.txt
cat $OUT/na_timing.txt
