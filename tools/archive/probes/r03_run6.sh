#!/bin/bash
# Round 3, GPU call 6: per-channel L2 write stalls of MANY pairs (fast and slow), instance-rotation A/B on slow and fast placements,
# f16 row-prefetch depth A/B.
set -o pipefail
out=gpurun_out/r03
mkdir -p $out/placement_pmc4
export TMPDIR=/tmp
PC_PAIRS=28 timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_EA0_WRREQ --kernel-trace --output-format json -d $out/placement_pmc4/raw -o p -- python3 tools/archive/probes/placement_counters.py > $out/placement_pmc4/raw.txt 2>&1 || { echo "raw pass failed"; tail -3 $out/placement_pmc4/raw.txt; }
ls -la $out/placement_pmc4/raw/ | tail -3
AB_DENSE=1 AB_ROUNDS=7 AB_TRIES=1 timeout -k 10 300 python tools/ab.py "" "MMDX_ROTATE=1" "MMDX_INTERLEAVE=0" "MMDX_INTERLEAVE=0 MMDX_ROTATE=1" > $out/rotate_ab_plain.txt 2>&1
cat $out/rotate_ab_plain.txt
AB_DENSE=1 AB_ROUNDS=7 AB_TRIES=1 timeout -k 10 300 python tools/ab.py "" "MMDX_ROTATE=1" > $out/rotate_ab_plain2.txt 2>&1
cat $out/rotate_ab_plain2.txt
AB_DENSE=1 AB_ROUNDS=7 timeout -k 10 300 python tools/ab.py "" "MMDX_ROTATE=1" > $out/rotate_ab_shopped.txt 2>&1
cat $out/rotate_ab_shopped.txt
V=build/variants
AB_WORKLOAD=c5x64 AB_ROUNDS=7 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py ra5=$V/libmmdx_ra5.so ra6=$V/libmmdx_ra6.so > $out/row_ahead_c5x64.txt 2>&1
cat $out/row_ahead_c5x64.txt
