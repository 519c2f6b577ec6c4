#!/bin/bash
# round 5, GPU session A: the new training tests with per-tensor error logging, the error-vs-rows table, step times.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT
( nproc; free -g; rocm-smi --showmeminfo vram 2>/dev/null | head -8 ) > $OUT/box.txt 2>&1
export GNN_TEST_ERRLOG=$OUT/errlog.jsonl
rm -f $GNN_TEST_ERRLOG
timeout 1500 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -s > $OUT/pytest_round5.log 2>&1; echo "round5 rc=$?" >> $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_dp.py -m gpu -q > $OUT/pytest_training.log 2>&1; echo "training rc=$?" >> $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fuzz.py -m gpu -q -k "training or train or gradients or composite_g or fuzz" > $OUT/pytest_r34_training.log 2>&1; echo "r34 rc=$?" >> $OUT/summary.txt
unset GNN_TEST_ERRLOG
python scripts/train_perf.py > $OUT/train_perf.txt 2>&1
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1
timeout 1500 python scripts/dev/train_error_vs_rows.py --out $OUT/r05_train_error_vs_rows.txt > $OUT/error_vs_rows.log 2>&1
tail -3 $OUT/pytest_round5.log $OUT/pytest_training.log $OUT/pytest_r34_training.log; cat $OUT/summary.txt $OUT/train_perf.txt $OUT/train_big.txt; cat $OUT/r05_train_error_vs_rows.txt
