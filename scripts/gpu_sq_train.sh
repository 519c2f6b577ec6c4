#!/bin/bash
# SQ / GRBM counters of the large-graph training kernels at C4 size: three rocprofv3 --pmc passes (kernel-trace only beside them) over
# scripts/train_big.py 1e6 1e7 64 10, summarised per kernel -> gpurun_out/$TAG/${TAG}_train_sq_counters.txt.  Run from the repo root on the GPU box:
#   bash scripts/gpu_sq_train.sh TAG
set -u
export TMPDIR=/tmp
TAG=${1:-r05h}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
pass() {   # name, counters...
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_$name -o sq -- python3 $ROOT/scripts/train_big.py 1e6 1e7 64 10 > $OUT/sq_$name.out 2> $OUT/sq_$name.err )
  f=$(find $OUT/sq_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|k_train_|k_aggregate_stats|k_aggregate_dz" $f > $OUT/sq_$name.csv
  rm -rf $OUT/sq_$name
}
pass a GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 - $OUT $TAG <<'PY'
import csv, sys, collections, os, re
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for name in 'abc':
    p = os.path.join(out, f'sq_{name}.csv')
    if not os.path.exists(p): continue
    seen = set()
    for r in csv.DictReader(open(p)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void gnn::', '').replace('gnn::', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        key = (k, r.get('Dispatch_Id'))
        if name == 'a' and key not in seen and r.get('Start_Timestamp'):
            seen.add(key); dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
lines = [f'# SQ / GRBM counters of the large-graph training kernels at C4 size (rocprofv3 --pmc, three passes over scripts/train_big.py 1e6 1e7 64 10, mean over the launches of',
         f'# 4 steps; {tag}).  clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration of the same pass (counter collection serialises the launches); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /',
         f'# (1024 SIMDs x GRBM_GUI_ACTIVE / 8); VALU issue = 4 cycles x (SQ_INSTS_VALU - SQ_INSTS_MFMA) over the same denominator.  A gfx950 SIMD issues either kind: the two fractions add.']
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    d = sum(dur[k]) / len(dur[k]) if dur[k] else float('nan')
    g = m.get('GRBM_GUI_ACTIVE', float('nan')) / 8
    den = 1024 * g
    valu = m.get('SQ_INSTS_VALU', float('nan')) - m.get('SQ_INSTS_MFMA', 0.0)
    lines.append(f"{k}: {d / 1e3:.1f} us under the counters, clock {g / d:.2f} GHz, MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / den:.2f}, VALU issue {4 * valu / den:.2f}, "
                 f"instructions per launch: VALU (without MFMA) {valu:.3g} MFMA {m.get('SQ_INSTS_MFMA', float('nan')):.3g} LDS {m.get('SQ_INSTS_LDS', float('nan')):.3g}; "
                 f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {m.get('SQ_WAIT_INST_ANY', float('nan')) / max(m.get('SQ_WAVE_CYCLES', float('nan')), 1):.2f}, LDS bank-conflict cycles / LDS active {m.get('SQ_LDS_BANK_CONFLICT', float('nan')) / max(m.get('SQ_LDS_IDX_ACTIVE', float('nan')), 1):.2f}")
    lines.append('   raw: ' + ', '.join(f'{c} {v:.0f}' for c, v in sorted(m.items())))
open(os.path.join(out, f'{tag}_train_sq_counters.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
