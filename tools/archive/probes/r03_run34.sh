#!/bin/bash
# Round 3, GPU call 50: instances per workgroup around 16 on a fast pair (tail of the last round of workgroups: 98 tiles x ceil(1024/g) workgroups over 768 slots).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
AB_TRIES=128 AB_WORKLOAD=c3 AB_ROUNDS=7 AB_ITERS=40 AB_PLAIN=0 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  g13=shipped:MMDX_GROUP=13 g14=shipped:MMDX_GROUP=14 g15=shipped:MMDX_GROUP=15 g16=shipped:MMDX_GROUP=16 g17=shipped:MMDX_GROUP=17 g18=shipped:MMDX_GROUP=18 g11=shipped:MMDX_GROUP=11 2>&1 | grep -v identical | tee $out/group_fine_sweep_fast_pair.txt
