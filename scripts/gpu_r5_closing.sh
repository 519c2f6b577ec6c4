#!/bin/bash
# the round's last GPU action: smoke, the whole -m gpu suite, then the closing evidence pass (scripts/gpu_r5_final.sh) on the same build
set -u
ROOT=$(pwd); TAG=${TAG:-r05h}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$? wall $(( $(date +%s) - t0 )) s"; tail -4 $OUT/pytest_gpu.log
TAG=$TAG bash scripts/gpu_r5_final.sh "round 5 closing pass" > $OUT/final.log 2>&1; tail -5 $OUT/final.log | cut -c1-400
