#!/bin/bash
# step time + kernel trace of the large-graph training step (no tests)
set -u
export TMPDIR=/tmp
ROOT=$(pwd); TAG=${TAG:-r05q}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1; tail -3 $OUT/train_big.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python3 $ROOT/scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_prof.out 2> $OUT/train_prof.err )
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_train_kernel_stats.csv && head -8 $OUT/${TAG}_train_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof_train
