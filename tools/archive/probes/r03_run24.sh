#!/bin/bash
# Round 3, GPU call 38: nt vs write-through on many plain placements of one process (which levels does write-through win at?).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
AB_TRIES=1 AB_PLAIN_N=9 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 600 python tools/archive/probes/store_policy_ab.py forced_nt=shipped:FLAGS=64 forced_wt=shipped:FLAGS=32 2>&1 | tee $out/store_policy_by_level.txt
