#!/bin/bash
# round-6 closing pass, all on the build that ships: smoke, the whole -m gpu suite, then the evidence of scripts/gpu_r5_final.sh (kernel-trace
# summaries + the two PMC passes of C3 / C4 / C5 / C4x4 / d = 200 -> profiles/hbm_traffic.json keyed to the library's source hash, the
# homogeneous training step's trace / PMC / SQ counters, the default bench line) plus the heterogeneous C5 training step's trace and PMC passes
# and the kernel trace of d = 128 on C4.
# Run from the repo root on the GPU box:  [TAG=r06] [SUITE=0] bash scripts/gpu_r6_closing.sh
set -u
export TMPDIR=/tmp
ROOT=$(pwd); TAG=${TAG:-r06}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
if [ "${SUITE:-1}" = 1 ]; then
  t0=$(date +%s)
  GNN_PARITY_OUT=$OUT/r06_train_c5_parity.txt timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$? wall $(( $(date +%s) - t0 )) s"; tail -4 $OUT/pytest_gpu.log
fi
# the C5 training step first: bench.py's training.c5_d64_k10 reads profiles/<tag>_c5_train_kernel_stats.json of THESE sources
bash scripts/gpu_profile_train_c5.sh ${TAG}_c5 > $OUT/train_c5_profile.log 2>&1 || true
python3 scripts/parse_kernel_stats.py $ROOT/gpurun_out/${TAG}_c5/${TAG}_c5_train_kernel_stats.csv 4 $OUT/r06_c5_train_kernel_stats.json >> $OUT/train_c5_profile.log 2>&1 \
  && cp $OUT/r06_c5_train_kernel_stats.json profiles/r06_c5_train_kernel_stats.json
bash scripts/gpu_pmc_train_c5.sh ${TAG}_c5 > $OUT/train_c5_pmc.log 2>&1 || true
COMMON="--warmup 1 --no-cpu-baseline --no-mutag --no-beyond-cache --no-training --steps 5"
trace() {   # name, bench args...: kernel-trace summary only
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python3 $ROOT/bench.py "$@" $COMMON > $OUT/bench_prof_$name.json 2> $OUT/bench_prof_$name.err )
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv && head -2 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-160
  rm -rf $OUT/prof_$name
}
trace d128 --workload c4 --state-dim 128 --max-iteration 20
TAG=$TAG bash scripts/gpu_r5_final.sh "round 6 closing pass" > $OUT/final.log 2>&1; tail -5 $OUT/final.log | cut -c1-600
