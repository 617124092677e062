#!/bin/bash
# Round 3, GPU call 22: ping-pong row buffers in the morph-row walk (tree) vs the rotating pair (variant pp0): parity + A/B.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
timeout -k 10 300 python tools/archive/probes/variant_check.py pp0=$V/libmmdx_pp0.so > $out/variant_check_pp.txt 2>&1; echo "variant check rc=$?"; grep -c "bit-exact" $out/variant_check_pp.txt; grep MISMATCH $out/variant_check_pp.txt
for wl in c3p c5x64 c2x64 c3; do
AB_WORKLOAD=$wl AB_ROUNDS=9 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py pp0=$V/libmmdx_pp0.so 2>&1 | tee -a $out/row_pingpong_ab.txt
done
timeout -k 10 200 python tools/fused_bench.py 2>&1 | tee $out/fused_bench_pp.txt
