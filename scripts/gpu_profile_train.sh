#!/bin/bash
# kernel-trace summary of one training step on a large graph (native gnn_train_step): which kernels the step spends its time in
set -u
export TMPDIR=/tmp
TAG=${1:-r02}; N=${2:-1e6}; E=${3:-1e7}; IT=${4:-10}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python3 $ROOT/scripts/train_big.py $N $E 64 $IT > $OUT/train_big.out 2> $OUT/train_big.err )
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 $f > $OUT/${TAG}_train_kernel_stats.csv && cut -c1-150 $OUT/${TAG}_train_kernel_stats.csv | head -24
rm -rf $OUT/prof_train
tail -6 $OUT/train_big.out
