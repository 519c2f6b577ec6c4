#!/bin/bash
# the large-graph training step after a kernel change: step time, its kernel trace, the large-graph parity tests and the error-vs-rows table
set -u
export TMPDIR=/tmp
ROOT=$(pwd); TAG=${TAG:-r05i}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1; cat $OUT/train_big.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python3 $ROOT/scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_prof.out 2> $OUT/train_prof.err )
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_train_kernel_stats.csv && head -12 $OUT/${TAG}_train_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof_train
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_training.py -m gpu -x -q -k "large_graph or thin_output_head or size_and_depth or full_constants or c4_size" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest.log
if [ "${ROWS_TABLE:-0}" = 1 ]; then timeout 1200 python scripts/dev/train_error_vs_rows.py > $OUT/error_vs_rows.txt 2>&1; tail -14 $OUT/error_vs_rows.txt; fi
