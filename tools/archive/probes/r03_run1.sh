#!/bin/bash
# Round 3, GPU call 1: regression tests of the advisor fixes + the store-policy / XCD-chunk / LDS-conflict A/Bs.
set -e -o pipefail
out=gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/gputest1.log 2>&1 || { tail -30 $out/gputest1.log; exit 1; }
tail -3 $out/gputest1.log
V=build/variants
timeout -k 10 400 python tools/archive/probes/store_policy_ab.py plain_copy=$V/libmmdx_sp0.so nt=$V/libmmdx_sp1.so sc1=$V/libmmdx_sp2.so sc0sc1=$V/libmmdx_sp3.so > $out/store_policy_c3.txt 2>&1
cat $out/store_policy_c3.txt
AB_WORKLOAD=v32 AB_ROUNDS=5 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py nt=$V/libmmdx_sp1.so sc1=$V/libmmdx_sp2.so > $out/store_policy_v32.txt 2>&1
cat $out/store_policy_v32.txt
AB_WORKLOAD=c3p AB_ROUNDS=5 AB_ITERS=20 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py nt=$V/libmmdx_sp1.so sc1=$V/libmmdx_sp2.so > $out/store_policy_c3p.txt 2>&1
cat $out/store_policy_c3p.txt
AB_WORKLOAD=c5x64 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py nt=$V/libmmdx_sp1.so sc1=$V/libmmdx_sp2.so > $out/store_policy_c5x64.txt 2>&1
cat $out/store_policy_c5x64.txt
FB_SWEEP="MMDX_XCD_CHUNK=0,1,2,4,8,12,64" timeout -k 10 400 python tools/fused_bench.py c2 c5 c3p --iters 10 > $out/xcd_chunk_sweep.txt 2>&1
cat $out/xcd_chunk_sweep.txt
timeout -k 10 300 python tools/archive/probes/lds_conflict_probe.py 16 12 > $out/lds_conflict_probe.txt 2>&1
cat $out/lds_conflict_probe.txt
