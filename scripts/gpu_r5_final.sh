#!/bin/bash
# round-5 closing evidence, all on the build that ships: kernel-trace summaries of C3 / C4 / C5 / C4x4 (4 M / 40 M) / d = 200, the two
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) of EVERY one of them -> profiles/hbm_traffic.json (every record keyed
# to gnnkeras_amd._native.source_hash()), the training step's trace, PMC passes and SQ counters, the default bench line.
# Run from the repo root on the GPU box:  [TAG=r05h] bash scripts/gpu_r5_final.sh [TAKEN-note]
set -u
export TMPDIR=/tmp
TAKEN=${1:-"round 5 closing pass"}
ROOT=$(pwd); TAG=${TAG:-r05h}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
echo '{"records": []}' > $OUT/hbm_traffic.json          # (records of other library sources are of no use: bench.py would refuse them)
COMMON="--warmup 1 --no-cpu-baseline --no-mutag --no-beyond-cache --no-training"
run() {   # name, bench args...
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python3 $ROOT/bench.py "$@" --steps 5 $COMMON > $OUT/bench_prof_$name.json 2> $OUT/bench_prof_$name.err )
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv && head -3 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-160
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -o $name -- python3 $ROOT/bench.py "$@" --steps 3 $COMMON > /dev/null 2> $OUT/pmc_${name}_$c.err )
    f=$(find $OUT/pmc_${name}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|k_state_" $f > $OUT/${TAG}_${name}_pmc_$c.csv
  done
  rm -rf $OUT/prof_$name $OUT/pmc_${name}_FETCH_SIZE $OUT/pmc_${name}_WRITE_SIZE
}
run c4 --workload c4
run c3 --workload c3
run c5 --workload c5
run c4x4 --workload c4x4
run d200 --workload c3 --nodes 3e5 --arcs 3e6 --state-dim 200 --max-iteration 20
P="python3 scripts/parse_pmc.py"
$P $OUT/${TAG}_c4_pmc_FETCH_SIZE.csv   $OUT/${TAG}_c4_pmc_WRITE_SIZE.csv   1e6 1e7 64 64  $OUT/hbm_traffic.json 128 k_state_fused "$TAKEN" >  $OUT/parse_pmc.log 2>&1
$P $OUT/${TAG}_c3_pmc_FETCH_SIZE.csv   $OUT/${TAG}_c3_pmc_WRITE_SIZE.csv   1e5 1e6 64 64  $OUT/hbm_traffic.json 256 k_state_fused "$TAKEN" >> $OUT/parse_pmc.log 2>&1
$P $OUT/${TAG}_c5_pmc_FETCH_SIZE.csv   $OUT/${TAG}_c5_pmc_WRITE_SIZE.csv   5e5 5e6 64 64  $OUT/hbm_traffic.json 256 k_state_fused "$TAKEN" >> $OUT/parse_pmc.log 2>&1
$P $OUT/${TAG}_c4x4_pmc_FETCH_SIZE.csv $OUT/${TAG}_c4x4_pmc_WRITE_SIZE.csv 4e6 4e7 64 64  $OUT/hbm_traffic.json 128 k_state_fused "$TAKEN" >> $OUT/parse_pmc.log 2>&1
$P $OUT/${TAG}_d200_pmc_FETCH_SIZE.csv $OUT/${TAG}_d200_pmc_WRITE_SIZE.csv 3e5 3e6 200 200 $OUT/hbm_traffic.json 800 k_state_xwide "$TAKEN" >> $OUT/parse_pmc.log 2>&1
cp $OUT/hbm_traffic.json profiles/hbm_traffic.json        # (so that the bench line below reports THIS pass's traffic)
bash scripts/gpu_profile_train.sh $TAG > $OUT/train_profile.log 2>&1 || true
bash scripts/gpu_pmc_train.sh $TAG > $OUT/train_pmc.log 2>&1 || true      # (FETCH / WRITE passes of the large-graph training kernels: ${TAG}_train_pmc.txt)
bash scripts/gpu_sq_train.sh $TAG > $OUT/train_sq.log 2>&1 || true         # (SQ / GRBM counters of the same kernels: ${TAG}_train_sq_counters.txt)
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 4000 $OUT/bench_default.json
python3 -c "from gnnkeras_amd._native import source_hash; print('library sources', source_hash())"
