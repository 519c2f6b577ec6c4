#!/bin/bash
# the shader clock a kernel actually ran at: rocprofv3 --pmc GRBM_GUI_ACTIVE (GPU-busy cycles at the shader clock) next to the
# kernel-trace durations, plus the matrix pipes' busy cycles (run from the repo root on the GPU box):
#   bash scripts/gpu_pmc_clock.sh TAG KERNEL_SUBSTRING script.py args...
set -u
export TMPDIR=/tmp
TAG=$1; KSUB=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/pmc_clk -o clk -- python3 $ROOT/"$@" > $OUT/pmc_clk.out 2> $OUT/pmc_clk.err )
f=$(find $OUT/pmc_clk -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" "$KSUB" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
acc = collections.defaultdict(list)
dur = []
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and 'Start_Timestamp' in r: dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
n = max(len(v) for v in acc.values()) if acc else 0
print('kernel', sys.argv[2], 'dispatches', n)
for k, v in sorted(acc.items()): print(f'{k:28s} mean {sum(v)/len(v):16.0f}')
if dur:
    d = sum(dur) / len(dur)
    print(f'duration mean {d/1e3:.1f} us -> GRBM_GUI_ACTIVE / duration = {sum(acc["GRBM_GUI_ACTIVE"])/len(acc["GRBM_GUI_ACTIVE"])/d*1e3:.0f} MHz')
PY
head -3 $f | cut -c1-400
rm -rf $OUT/pmc_clk
