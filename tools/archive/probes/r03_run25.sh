#!/bin/bash
# Round 3, GPU call 39: workgroup shape under write-through stores on plainly allocated (not fast) arrays.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  wt_g8=shipped:FLAGS=32,MMDX_GROUP=8 wt_g12=shipped:FLAGS=32,MMDX_GROUP=12 wt_g16=shipped:FLAGS=32,MMDX_GROUP=16 \
  wt_g24=shipped:FLAGS=32,MMDX_GROUP=24 wt_g32=shipped:FLAGS=32,MMDX_GROUP=32 wt_blocked=shipped:FLAGS=32,MMDX_INTERLEAVE=0 \
  wt_512thr=shipped:FLAGS=32,MMDX_THREADS=512 2>&1 | tee $out/shape_sweep_write_through.txt
