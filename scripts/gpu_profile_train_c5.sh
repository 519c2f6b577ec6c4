#!/bin/bash
# kernel-trace summary of training steps of a heterogeneous model on a large graph (BASELINE C5 by default): which kernels the step spends its time in
set -u
export TMPDIR=/tmp
TAG=${1:-r06_c5}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python3 $ROOT/scripts/train_c5.py "$@" > $OUT/train_c5.out 2> $OUT/train_c5.err )
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 $f > $OUT/${TAG}_train_kernel_stats.csv && cut -c1-170 $OUT/${TAG}_train_kernel_stats.csv | head -36
rm -rf $OUT/prof_train
tail -7 $OUT/train_c5.out
