#!/bin/bash
# Round 3, GPU call 34: the same A/B (nt vs sc1 nt through the builtin) on a box that offers a fast placement; SoA only, plus f16 positions.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
AB_TRIES=128 AB_WORKLOAD=c3 AB_ROUNDS=9 AB_ITERS=40 AB_PLAIN=1 timeout -k 10 400 python tools/archive/probes/store_policy_ab.py sc1nt=$V/libmmdx_sp6.so 2>&1 | tee -a $out/store_policy_buffer_builtin_ab2.txt
AB_TRIES=128 AB_WORKLOAD=v32 AB_ROUNDS=7 AB_ITERS=40 AB_PLAIN=1 timeout -k 10 400 python tools/archive/probes/store_policy_ab.py sc1nt=$V/libmmdx_sp6.so 2>&1 | tee -a $out/store_policy_buffer_builtin_ab2.txt
