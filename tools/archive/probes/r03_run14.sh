#!/bin/bash
# Round 3, GPU call 24: shared morph pass in latency order (variant mp1) vs the shipped pass: parity + whole-step A/B.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
timeout -k 10 300 python tools/archive/probes/variant_check.py mp1=$V/libmmdx_mp1.so > $out/variant_check_mp.txt 2>&1; echo "variant check rc=$?"; grep -c "bit-exact" $out/variant_check_mp.txt; grep MISMATCH $out/variant_check_mp.txt
for wl in c3 v32; do
AB_WORKLOAD=$wl AB_ROUNDS=11 AB_ITERS=40 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py mp1=$V/libmmdx_mp1.so 2>&1 | tee -a $out/morph_pass_latency_order_ab.txt
done
