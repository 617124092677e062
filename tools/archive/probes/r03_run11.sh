#!/bin/bash
# Round 3, GPU call 20: which SIMD a workgroup's waves land on; wave <-> sorted-block pairing A/B (MMDX_WAVE_BLOCKS 1 / 2) + parity.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 60 tools/archive/probes/simd_map_probe > $out/simd_map_probe.txt 2>&1; echo "simd probe rc=$?"; cat $out/simd_map_probe.txt
V=build/variants
timeout -k 10 300 python tools/archive/probes/variant_check.py wb1=$V/libmmdx_wb1.so wb2=$V/libmmdx_wb2.so > $out/variant_check_wb.txt 2>&1; echo "variant check rc=$?"; grep -c "bit-exact" $out/variant_check_wb.txt; grep MISMATCH $out/variant_check_wb.txt
for wl in c3 c3p c5x64 c2x64; do
AB_WORKLOAD=$wl AB_ROUNDS=9 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py wb1=$V/libmmdx_wb1.so wb2=$V/libmmdx_wb2.so 2>&1 | tee -a $out/wave_blocks_ab.txt
done
