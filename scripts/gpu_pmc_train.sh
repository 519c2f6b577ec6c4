#!/bin/bash
# HBM traffic of the large-graph training kernels: the two PMC passes (FETCH_SIZE, WRITE_SIZE - each alone with --kernel-trace, as
# MI355X_MICROARCH.md prescribes) over scripts/train_big.py at C4 size, then per-kernel means next to the algorithmic bytes.
set -u
export TMPDIR=/tmp
TAG=${1:-r03p}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o t -- python3 $ROOT/scripts/train_big.py 1e6 1e7 64 10 > $OUT/pmc_$c.out 2> $OUT/pmc_$c.err )
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|k_train_|k_aggregate_stats|k_aggregate_vec|k_aggregate_dz|k_rows_stats|k_stats_finish" $f > $OUT/${TAG}_train_pmc_$c.csv
  rm -rf $OUT/pmc_$c
done
python3 - $OUT/${TAG}_train_pmc_FETCH_SIZE.csv $OUT/${TAG}_train_pmc_WRITE_SIZE.csv <<'PY' | tee $OUT/${TAG}_train_pmc.txt
import csv, sys, collections
def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter: acc[r['Kernel_Name'].replace('void gnn::', '').split('(')[0]].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
f, w = means(sys.argv[1], 'FETCH_SIZE'), means(sys.argv[2], 'WRITE_SIZE')
N, E, S = 1e6, 1e7, 64
import os
dz = os.environ.get('GNN_TRAIN_DZ', '1') != '0'       # round 5: the dense kernels read dZ alone (no G + Y), k_train_bwd_dx skips the state rows
alg = {'k_aggregate_stats': E * (4 + 4 * S) + N * (4 + 4 * S), 'k_aggregate_vec': E * (4 + 4 * S) + N * (4 + 3 * 4 * S),
       'k_aggregate_dz': E * (4 + 4 * S) + N * (4 + 4 * 4 * S),      # gathered dx_agg rows + source ids; per node row pointer, dx_state', state_t, dZ out
       'k_train_fwd': N * (4 * S * 3 + 128), 'k_train_wgrad': N * (4 * S * (3 if dz else 4) + 128),
       'k_train_bwd_dx': N * (4 * S * (2 if dz else 4) + 8 * S)}     # (substring match: also k_train_fwd_b6, k_train_wgrad32, k_train_bwd_dx_b6)
print('# HBM bytes per launch from the PMC counters (KiB units; FETCH doubled: gfx950 tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes)')
print('# kernel, launches, FETCH raw MB, 2 x FETCH + WRITE MB, algorithmic MB, ratio')
for k in sorted(f):
    fr, n = f[k]; wr = w.get(k, (0.0, 0))[0]
    tot = (2 * fr + wr) * 1024 / 1e6
    a = next((v for name, v in alg.items() if name in k), None)
    print(f'{k:48s} {n:4d} {fr * 1024 / 1e6:9.1f} {tot:9.1f} ' + (f'{a / 1e6:9.1f} {tot / (a / 1e6):5.2f}' if a else '        -     -'))
PY
