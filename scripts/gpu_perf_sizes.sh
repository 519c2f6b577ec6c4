#!/bin/bash
# per-iteration time of the fused iteration at the BASELINE sizes and in between (run from the repo root on the GPU box)
set -u
OUT=gpurun_out/${1:-perf}; mkdir -p $OUT
for cfg in "3e4 3e5 64" "1e5 1e6 64" "3e5 3e6 64" "1e6 1e7 64" "1e6 1e7 32" "1e5 1e6 32"; do
  set -- $cfg
  timeout 600 python scripts/er_perf.py $1 $2 $3 2>&1 | tail -1 | sed "s/^/N=$1 E=$2 d=$3: /" | tee -a $OUT/er_sizes.txt
done
timeout 600 python scripts/c5_perf.py 2>&1 | tail -2 | tee -a $OUT/er_sizes.txt
