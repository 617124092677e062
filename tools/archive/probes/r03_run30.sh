#!/bin/bash
# Round 3, GPU call 46: write-through for the f16-position layout (256 instances of the 256k-vertex model, shared morphs) on plain arrays.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=c5s AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  forced_nt=shipped:FLAGS=64 forced_wt=shipped:FLAGS=32 wt_g16=shipped:FLAGS=32,MMDX_GROUP=16 nt_g8=shipped:FLAGS=64,MMDX_GROUP=8 2>&1 | grep -v identical | tee $out/store_policy_pos16_crowd.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
