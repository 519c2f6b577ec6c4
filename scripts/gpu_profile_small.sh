#!/bin/bash
# kernel-trace summaries of the small / mid-size paths: MUTAG with grouped launches, a 30 k-node graph (mid-size whole-loop
# kernel)  (run from the repo root on the GPU box)
set -u
export TMPDIR=/tmp
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
prof() {  # name, program args...
  name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python3 "$@" > $OUT/prof_$name.out 2> $OUT/prof_$name.err )
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -14 $f > $OUT/${TAG}_${name}_kernel_stats.csv && head -5 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-170
  rm -rf $OUT/prof_$name
}
prof mutag_groups $ROOT/scripts/mutag_groups.py
prof n30k_mid $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mutag --no-beyond-cache --workload c3 --nodes 3e4 --arcs 3e5
tail -4 $OUT/prof_mutag_groups.out
