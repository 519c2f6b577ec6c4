#!/bin/bash
# kernel-trace summary of one script:  bash scripts/gpu_profile_script.sh TAG NAME script.py args...   -> gpurun_out/TAG/TAG_NAME_kernel_stats.csv
set -u
export TMPDIR=/tmp
TAG=$1; NAME=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$NAME -o $NAME -- python3 $ROOT/"$@" > $OUT/$NAME.out 2> $OUT/$NAME.err )
f=$(find $OUT/prof_$NAME -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 $f > $OUT/${TAG}_${NAME}_kernel_stats.csv && cut -c1-170 $OUT/${TAG}_${NAME}_kernel_stats.csv | head -8
rm -rf $OUT/prof_$NAME
tail -3 $OUT/$NAME.out
