#!/bin/bash
# kernel-trace summaries of the secondary shapes: C3, a 30 k-node graph, d = 128 at C4 size (run from the repo root on the GPU box)
set -u
export TMPDIR=/tmp
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
run() {  # name, bench args...
  name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mutag --no-beyond-cache "$@" > $OUT/bench_prof_$name.json 2> $OUT/bench_prof_$name.err )
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 $f > $OUT/${TAG}_${name}_kernel_stats.csv && head -3 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-160
  rm -rf $OUT/prof_$name
}
run c3 --workload c3
run n30k --workload c3 --nodes 3e4 --arcs 3e5
run c4_d128 --workload c4 --state-dim 128
run c4_d32 --workload c4 --state-dim 32
