#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT
python scripts/train_big.py 1e6 1e7 64 10 > $OUT/train_big.txt 2>&1; cat $OUT/train_big.txt
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -x -q -k "large_graph or thin_output_head or size_and_depth or full_constants" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest.log
