#!/bin/bash
# Round 3, GPU call 45: smaller crowds (128 / 256 / 512 instances of the 50k model) on plain arrays: nt vs write-through (+ 8 per workgroup).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
for ni in 128 256 512; do
echo "#### $ni instances" | tee -a $out/store_policy_small_crowds.txt
AB_NI=$ni AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=60 AB_PLAIN=1 timeout -k 10 600 python tools/archive/probes/store_policy_ab.py \
  forced_nt=shipped:FLAGS=64 forced_wt=shipped:FLAGS=32 2>&1 | grep -v identical | tee -a $out/store_policy_small_crowds.txt
done
