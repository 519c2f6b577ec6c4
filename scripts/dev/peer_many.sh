#!/bin/bash
# the peer-store exchange between W processes on ONE GPU (tests/peer_worker.py is a rank; the suite runs W = 2): bash scripts/dev/peer_many.sh W [d] [thr]
set -u
W=${1:-3}; D=${2:-64}; THR=${3:-0.0}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$(( 20000 + RANDOM % 20000 )) WORLD_SIZE=$W PEER_D=$D PEER_THR=$THR HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$(pwd)
pids=()
for r in $(seq 0 $(( W - 1 ))); do RANK=$r timeout 600 python3 tests/peer_worker.py > /tmp/peer_$r.log 2>&1 & pids+=($!); done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
tail -3 /tmp/peer_0.log; echo "W=$W d=$D thr=$THR rc=$rc"
