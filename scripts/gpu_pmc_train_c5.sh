#!/bin/bash
# Fabric traffic of the heterogeneous large-graph training kernels (BASELINE C5): the two PMC passes (FETCH_SIZE, WRITE_SIZE - each alone with
# --kernel-trace, as MI355X_MICROARCH.md prescribes) over scripts/train_c5.py, then per-kernel means next to the algorithmic bytes per launch.
set -u
export TMPDIR=/tmp
TAG=${1:-r06_c5}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o t -- python3 $ROOT/scripts/train_c5.py > $OUT/pmc_$c.out 2> $OUT/pmc_$c.err )
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|k_train_|k_aggregate_stats|k_aggregate_dz|k_head_" $f > $OUT/${TAG}_train_pmc_$c.csv
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
N, E, S, T, XW = 5e5, 5e6, 64, 3, 64
# per LAUNCH: the aggregates run once per node type (a third of the rows and arcs each), the dense kernels once for all types
alg = {'k_aggregate_stats': (E * (4 + 4 * S) + N * (4 + 4 * S)) / T,
       'k_aggregate_dz': (E * (4 + 4 * S) + N * (4 + 4 * 4 * S)) / T,     # gathered dx_agg rows + source ids; per row: pointer, dx_state', state_t, dZ out
       'k_train_fwd_b6_types': N * (4 * S * 4),                           # state, agg, the constant part Cc in; the new state out
       'k_train_wgrad_b6_types': N * (4 * S * 3 + 4 * XW),                # dZ, state, agg, the constants line
       'k_train_bwd_dx_b6_types': N * (4 * S * 2 + 8 * S)}                # dZ, agg in; dx [state | agg] out
print('# L2 -> fabric bytes per launch from the PMC counters (KiB units; FETCH doubled: gfx950 tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes;')
print('# Infinity-Cache hits included - this is not an HBM-only figure)')
print('# kernel, launches, FETCH raw MB, 2 x FETCH + WRITE MB, algorithmic MB, ratio')
for k in sorted(f):
    fr, n = f[k]; wr = w.get(k, (0.0, 0))[0]
    tot = (2 * fr + wr) * 1024 / 1e6
    a = next((v for name, v in alg.items() if name in k), None)
    print(f'{k:52s} {n:4d} {fr * 1024 / 1e6:9.1f} {tot:9.1f} ' + (f'{a / 1e6:9.1f} {tot / (a / 1e6):5.2f}' if a else '        -     -'))
PY
