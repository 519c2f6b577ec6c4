#!/bin/bash
# round 3: one rank's share of an R-GPU C4 run on ONE GPU (bench.py --emulate-shard r/R): JSON line + rocprofv3 kernel stats
set -u
export TMPDIR=/tmp
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for R in 2 4 8; do
  python bench.py --emulate-shard 1/$R --steps 20 > $OUT/shard_1of${R}.json 2> $OUT/shard_1of${R}.err
  python bench.py --emulate-shard 1/$R --steps 20 --no-overlap > $OUT/shard_1of${R}_nooverlap.json 2>> $OUT/shard_1of${R}.err
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_s$R -o s$R -- python3 $ROOT/bench.py --emulate-shard 1/$R --steps 20 > /dev/null 2> $OUT/prof_s$R.err )
  f=$(find $OUT/prof_s$R -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_shard_1of${R}_kernel_stats.csv && head -5 $OUT/${TAG}_shard_1of${R}_kernel_stats.csv
  cat $OUT/shard_1of${R}.json | head -c 1500; echo
done
rm -rf $OUT/prof_s*
