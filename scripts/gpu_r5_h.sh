#!/bin/bash
set -u
ROOT=$(pwd); TAG=${TAG:-r05j}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_training.py -m gpu -x -q --durations=10 > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -16 $OUT/pytest.log
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1; tail -4 $OUT/train_big.txt
