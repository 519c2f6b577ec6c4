#!/bin/bash
set -u
OUT=gpurun_out/${1:-wide}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide_state or hub_rows" > $OUT/pytest_wide.log 2>&1; tail -5 $OUT/pytest_wide.log
timeout 600 python scripts/er_perf.py 1e6 1e7 128 > $OUT/er_perf_128.txt 2>&1; tail -2 $OUT/er_perf_128.txt
timeout 600 python scripts/er_perf.py 1e6 1e7 96 > $OUT/er_perf_96.txt 2>&1; tail -1 $OUT/er_perf_96.txt
GNN_UNFUSED=1 timeout 600 python scripts/er_perf.py 1e5 1e6 128 > $OUT/er_perf_128_c3.txt 2>&1; tail -1 $OUT/er_perf_128_c3.txt
