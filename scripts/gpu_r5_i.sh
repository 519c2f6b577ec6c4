#!/bin/bash
set -u
ROOT=$(pwd); TAG=${TAG:-r05m}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1; tail -4 $OUT/train_big.txt
t0=$(date +%s)
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$? wall $(( $(date +%s) - t0 )) s"; tail -22 $OUT/pytest_gpu.log
