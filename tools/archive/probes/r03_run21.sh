#!/bin/bash
# Round 3, GPU call 33: write-through stores through the buffer-store builtin (6: sc1 nt, 7: sc1) vs shipped nt, on shopped and
# plainly allocated arrays; parity of the variants.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
timeout -k 10 300 python tools/archive/probes/variant_check.py sp6=$V/libmmdx_sp6.so sp7=$V/libmmdx_sp7.so > $out/variant_check_wt.txt 2>&1; echo "variant check rc=$?"; grep -c "bit-exact" $out/variant_check_wt.txt; grep MISMATCH $out/variant_check_wt.txt
for wl in c3 v32; do
AB_WORKLOAD=$wl AB_ROUNDS=9 AB_ITERS=40 AB_PLAIN=1 timeout -k 10 400 python tools/archive/probes/store_policy_ab.py sc1nt=$V/libmmdx_sp6.so sc1=$V/libmmdx_sp7.so 2>&1 | tee -a $out/store_policy_buffer_builtin_ab.txt
done
