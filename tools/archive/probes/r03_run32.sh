#!/bin/bash
# Round 3, GPU call 48: write-through crowd kernel held to 96 / 80 VGPRs (five / six waves per SIMD) with 6 or 4 instances per workgroup.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
AB_TRIES=1 AB_PLAIN_N=2 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  w5_g6=$V/libmmdx_wtw5.so:FLAGS=32,MMDX_GROUP=6 w5_g8=$V/libmmdx_wtw5b.so:FLAGS=32,MMDX_GROUP=8 w6_g4=$V/libmmdx_wtw6.so:FLAGS=32,MMDX_GROUP=4 2>&1 | grep -v identical | tee $out/write_through_more_waves.txt
