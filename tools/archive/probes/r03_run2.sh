#!/bin/bash
# Round 3, GPU call 2: nt stores shipped -- parity of the variants, the remaining store policies, launch-shape re-sweep under nt,
# LDS-conflict counters of the conflict-free model, creation modes, and a first default bench line.
set -e -o pipefail
out=gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
V=build/variants
timeout -k 10 400 python tools/archive/probes/variant_check.py plain=$V/libmmdx_sp0.so sc1nt=$V/libmmdx_sp4.so > $out/variant_check.txt 2>&1 || { cat $out/variant_check.txt; echo "VARIANT CHECK FAILED"; }
cat $out/variant_check.txt
timeout -k 10 400 python tools/archive/probes/store_policy_ab.py plain=$V/libmmdx_sp0.so sc1nt=$V/libmmdx_sp4.so sc0nt=$V/libmmdx_sp5.so > $out/store_policy_c3_b.txt 2>&1
cat $out/store_policy_c3_b.txt
AB_DENSE=1 AB_ROUNDS=7 timeout -k 10 400 python tools/ab.py "" "MMDX_GROUP=12" "MMDX_GROUP=20" "MMDX_GROUP=24" "MMDX_GROUP=32" "MMDX_GROUP=8" "MMDX_THREADS=512" "MMDX_INTERLEAVE=0" > $out/shape_sweep_nt.txt 2>&1
cat $out/shape_sweep_nt.txt
AB_ROUNDS=5 timeout -k 10 400 python tools/mode_ab.py > $out/creation_modes_nt.txt 2>&1
cat $out/creation_modes_nt.txt
for w in 16 12; do
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/lds_w$w -o lds -- python3 tools/archive/probes/lds_conflict_probe.py $w > $out/lds_w$w.txt 2>&1
done
timeout -k 10 500 python bench.py > $out/bench_nt_first.json 2> $out/bench_nt_first.err
cat $out/bench_nt_first.json
