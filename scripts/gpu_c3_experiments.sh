#!/bin/bash
# the C3 variants of the wave-specialised kernel, measured (experiment library: make -C gnnkeras_amd/csrc c3exp)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05c3; mkdir -p $OUT
export GNNKERAS_AMD_LIB=$ROOT/gnnkeras_amd/csrc/libgnnloop_c3exp.so
{
echo "# C3 (100 k nodes / 1 M arcs, d = 64, 'average') on the experiment build of k_state_fused4 (-DGNN_F4_EXPERIMENT -DGNN_F4_TIMELINE: the time stamps cost ~1 us per launch);"
echo "# variant 0 = the shipping kernel (16 lanes x 16 B per row, 4 rows in flight), 1 = + first-job header, 2 = 8 lanes x 32 B per row at 2 rows in flight,"
echo "# 3 = 2 + header, 4 = 8 lanes x 32 B at 4 rows in flight (76 B of scratch), 5 = 4 + header, 8 = shipping lanes at 2 rows in flight"
for v in 0 1 2 3 4 5 8 0; do GNN_F4_VARIANT=$v python scripts/c3_variant.py 1e5; done
echo "# the same variants at 200 k / 2 M and at C4 size (1 M / 10 M; the C form there: GNN_XC=0)"
for v in 0 1 2 3; do GNN_F4_VARIANT=$v python scripts/c3_variant.py 2e5; done
for v in 0 1 2 3; do GNN_XC=0 GNN_F4_VARIANT=$v python scripts/c3_variant.py 1e6; done
} > $OUT/r05_c3_experiments.txt 2> $OUT/err.txt
cat $OUT/r05_c3_experiments.txt; tail -5 $OUT/err.txt
