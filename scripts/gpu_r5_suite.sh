#!/bin/bash
# the whole -m gpu suite as the driver runs it (one process, -x), with its wall time and the slowest tests
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05suite; mkdir -p $OUT
export GNN_TEST_ERRLOG=$OUT/errlog.jsonl; rm -f $GNN_TEST_ERRLOG
T0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=25 > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "suite rc=$? wall $(( $(date +%s) - T0 )) s" > $OUT/summary.txt
tail -45 $OUT/pytest_gpu.log; cat $OUT/summary.txt
python bench.py --force-sharded --workload c3 --steps 3 --warmup 1 --no-mutag > $OUT/bench_forced_interp.json 2> $OUT/bench_forced_interp.err
python bench.py --force-sharded --workload c3 --steps 3 --warmup 1 --no-mutag --native-loop > $OUT/bench_forced_native.json 2> $OUT/bench_forced_native.err
for extra in "" "--native-loop" "--pipeline-chunks 4" "--pipeline-chunks 4 --native-loop"; do
  python bench.py --emulate-shard 0/8 $extra > $OUT/shard_0of8_$(echo $extra | tr -d ' -').json 2>> $OUT/shard.err
done
for f in $OUT/bench_forced_*.json; do python -c "import json; r=json.loads([l for l in open('$f') if l.startswith('{')][0]); print('$f', r.get('loop_driver'), r.get('host_issue_us_per_iteration'), r.get('per_iteration_ms'), r['value'])"; done
for f in $OUT/shard_0of8_*.json; do python -c "import json; r=json.load(open('$f')); print(r['loop_driver'], r['pipeline_chunks'], 'host us/iter %.1f' % r['host_issue_us_per_iteration'], 'kernel ms', r['per_iteration_ms']['kernel'])"; done
