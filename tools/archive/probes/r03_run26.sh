#!/bin/bash
# Round 3, GPU call 40: small groups under write-through (and nt) stores on arrays that are not fast.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  nt_g8=shipped:FLAGS=64,MMDX_GROUP=8 nt_g4=shipped:FLAGS=64,MMDX_GROUP=4 wt_g4=shipped:FLAGS=32,MMDX_GROUP=4 wt_g6=shipped:FLAGS=32,MMDX_GROUP=6 wt_g8=shipped:FLAGS=32,MMDX_GROUP=8 \
  wt_g10=shipped:FLAGS=32,MMDX_GROUP=10 wt_g8_blocked=shipped:FLAGS=32,MMDX_GROUP=8,MMDX_INTERLEAVE=0 wt_g16=shipped:FLAGS=32,MMDX_GROUP=16 2>&1 | tee $out/shape_sweep_write_through2.txt
