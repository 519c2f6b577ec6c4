#!/bin/bash
# one GPU-box visit: the -m gpu suite and the default bench line (run from the repo root)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-run}
mkdir -p $OUT
python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
if [ -z "${NO_BENCH:-}" ]; then
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  tail -c 600 $OUT/bench.json
fi
