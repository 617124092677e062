#!/bin/bash
# Round 3, GPU call 10: GPU test suite of the current tree; z-pairing in the per-instance-morph walk (variant build) A/B + parity.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/gputest3.log 2>&1 || { tail -30 $out/gputest3.log; exit 1; }
tail -2 $out/gputest3.log
V=build/variants
timeout -k 10 300 python tools/archive/probes/variant_check.py zpair=$V/libmmdx_zp1.so > $out/variant_check_zpair.txt 2>&1; echo "variant check rc=$?"; grep -c "bit-exact" $out/variant_check_zpair.txt; grep MISMATCH $out/variant_check_zpair.txt
for wl in c3p c5x64 c2x64; do
AB_WORKLOAD=$wl AB_ROUNDS=9 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py zpair=$V/libmmdx_zp1.so shipped_copy=$V/libmmdx_shipcopy.so 2>&1 | tee -a $out/zpair_ab.txt
done
