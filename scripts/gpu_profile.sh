#!/bin/bash
# rocprofv3 summaries of bench.py workloads (run from the repo root on the GPU box): kernel-trace stats per workload, and the
# two PMC passes (FETCH_SIZE / WRITE_SIZE, each alone with --kernel-trace) for the workloads named in $PMC
set -u
export TMPDIR=/tmp
TAG=${1:-r02}; shift || true
WORKLOADS=${WORKLOADS:-"c4 c5 c4x4"}
PMC=${PMC:-"c4"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for w in $WORKLOADS; do
  steps=5; [ $w = c4x4 ] && steps=2
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python3 $ROOT/bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-mutag --no-beyond-cache --no-training "$@" > $OUT/bench_prof_$w.json 2> $OUT/bench_prof_$w.err )
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${w}_kernel_stats.csv && head -4 $OUT/${TAG}_${w}_kernel_stats.csv
done
for w in $PMC; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o $w -- python3 $ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-mutag --no-beyond-cache --no-training "$@" > /dev/null 2> $OUT/pmc_${w}_$c.err )
    f=$(find $OUT/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|k_state_fused" $f > $OUT/${TAG}_${w}_pmc_$c.csv
  done
done
rm -rf $OUT/prof_* $OUT/pmc_*_FETCH_SIZE $OUT/pmc_*_WRITE_SIZE
ls -la $OUT
