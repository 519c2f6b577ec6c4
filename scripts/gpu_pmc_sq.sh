#!/bin/bash
# one rocprofv3 --pmc pass of SQ wave-state counters over a script (run from the repo root on the GPU box):
#   bash scripts/gpu_pmc_sq.sh TAG KERNEL_SUBSTRING script.py args...
set -u
export TMPDIR=/tmp
TAG=$1; KSUB=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/pmc_sq -o sq -- python3 $ROOT/"$@" > $OUT/pmc_sq.out 2> $OUT/pmc_sq.err )
f=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" "$KSUB" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
acc = collections.defaultdict(list)
for r in rows: acc[r['Counter_Name']].append(float(r['Counter_Value']))
n = max(len(v) for v in acc.values()) if acc else 0
print('kernel', sys.argv[2], 'dispatches', n)
for k, v in sorted(acc.items()): print(f'{k:28s} mean {sum(v)/len(v):16.0f}')
PY
rm -rf $OUT/pmc_sq
