#!/bin/bash
# round-3 closing evidence: kernel-trace summaries of C3 / C4 / C4x4 / C5 on the final build, the two PMC passes of C4, the default bench line
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03f; mkdir -p $OUT
WORKLOADS="c3 c4 c5 c4x4" PMC="c4" bash scripts/gpu_profile.sh r03f > $OUT/profile.log 2>&1
python3 scripts/parse_pmc.py $OUT/r03f_c4_pmc_FETCH_SIZE.csv $OUT/r03f_c4_pmc_WRITE_SIZE.csv 1e6 1e7 64 64 $OUT/hbm_traffic.json 128 > $OUT/parse_pmc.log 2>&1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json
