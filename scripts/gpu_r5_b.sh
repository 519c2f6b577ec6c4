#!/bin/bash
# round 5, GPU session B: heterogeneous models on the persistent small-graph kernels, the native shard loop, tightened bars.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT
export GNN_TEST_ERRLOG=$OUT/errlog.jsonl
rm -f $GNN_TEST_ERRLOG
timeout 1500 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -s -k "not size_and_depth" > $OUT/pytest_round5.log 2>&1; echo "round5 rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest_multi.log 2>&1; echo "multi rc=$?" >> $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_dp.py tests/test_gpu_fuzz.py -m gpu -q > $OUT/pytest_training.log 2>&1; echo "training rc=$?" >> $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -k "training or train or gradients or composite or persistent or tiles" > $OUT/pytest_r34_training.log 2>&1; echo "r34 rc=$?" >> $OUT/summary.txt
unset GNN_TEST_ERRLOG
python scripts/train_perf.py > $OUT/train_perf.txt 2>&1
python - > $OUT/composite_perf.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.composite_training_section(torch.device('cuda', 0))))
PY
for extra in "" "--native-loop" "--pipeline-chunks 4" "--pipeline-chunks 4 --native-loop"; do
  python bench.py --emulate-shard 0/8 $extra > $OUT/shard_0of8_$(echo $extra | tr -d ' -').json 2>> $OUT/shard.err
done
tail -n 4 $OUT/pytest_round5.log $OUT/pytest_multi.log $OUT/pytest_training.log $OUT/pytest_r34_training.log; cat $OUT/summary.txt $OUT/train_perf.txt $OUT/composite_perf.txt
for f in $OUT/shard_0of8_*.json; do echo $f; python -c "import json,sys; r=json.load(open('$f')); print(r['loop_driver'], r['pipeline_chunks'], 'host us/iter %.1f' % r['host_issue_us_per_iteration'], 'kernel ms', r['per_iteration_ms']['kernel'])"; done
