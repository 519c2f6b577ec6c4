#!/bin/bash
# flakiness check: the tests that run the kernels with hand-counted waits (LDS rings) and the buffered gathers, several times over
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05rep; mkdir -p $OUT
for i in 1 2 3; do
  timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_training.py -m gpu -x -q -p no:cacheprovider \
      -k "large_graph or ragged_last_tile or takes_both_gradients or thin_output_head or c4_size or size_and_depth" > $OUT/rep_$i.log 2>&1
  echo "repetition $i rc=$? $(tail -1 $OUT/rep_$i.log)"
done
