#!/bin/bash
# Round 3, GPU call 30: per-instance-morph kernel forced to 80 VGPRs (three 8-wave workgroups per CU, 157 spills) vs shipped.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
for wl in c3p c5x64 c2x64; do
AB_WORKLOAD=$wl AB_ROUNDS=7 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py f4w6_lds48k=$V/libmmdx_f4w6.so:MMDX_LDS_TARGET=49152 2>&1 | tee -a $out/fused4_six_waves_ab.txt
done
