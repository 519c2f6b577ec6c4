#!/bin/bash
# kernel-trace summary of 2 x 20 in-library training steps on MUTAG batches (d = 32, 50 iterations): persistent small-graph kernels
set -u
export TMPDIR=/tmp
TAG=${1:-r03}; D=${2:-32}; IT=${3:-50}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ts -o ts -- python3 $ROOT/scripts/train_profile_native.py $D $IT > $OUT/train_small.out 2> $OUT/train_small.err )
f=$(find $OUT/prof_ts -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 $f > $OUT/${TAG}_train_small_kernel_stats.csv && cut -c1-160 $OUT/${TAG}_train_small_kernel_stats.csv | head -40
rm -rf $OUT/prof_ts
tail -3 $OUT/train_small.out
