#!/bin/bash
# round 5, GPU session C: the dZ flow of the large-graph backward sweep (k_aggregate_dz), step time, then the whole suite
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05c; mkdir -p $OUT
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1
GNN_TRAIN_DZ=0 python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big_dz0.txt 2>&1
cat $OUT/train_big.txt $OUT/train_big_dz0.txt
export GNN_TEST_ERRLOG=$OUT/errlog.jsonl; rm -f $GNN_TEST_ERRLOG
T0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=25 > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "suite rc=$? wall $(( $(date +%s) - T0 )) s" > $OUT/summary.txt
tail -45 $OUT/pytest_gpu.log; cat $OUT/summary.txt
