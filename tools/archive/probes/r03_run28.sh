#!/bin/bash
# Round 3, GPU call 43: other cache-policy bits of the write-through flavour (sc0 sc1 nt / sc0 nt / sc0 sc1) on arrays that are not fast.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  sc0_sc1_nt=$V/libmmdx_aux19.so sc0_nt=$V/libmmdx_aux3.so sc0_sc1=$V/libmmdx_aux17.so 2>&1 | tee $out/write_through_bits_ab.txt
