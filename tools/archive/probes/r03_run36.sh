#!/bin/bash
# Round 3, GPU call 53: long soak of the final tree (one run): 15 000 randomized models x every call form, the same 10 000 with
# write-through forced, 8 000 rigs x 128 instances.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
f=$out/soak_long_final.txt
echo "== 15000 seeds of randomized models x every call form" > $f
MMDX_SOAK_SEEDS=15000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -1 >> $f
echo "== MMDX_STORE_WT=1, 10000 seeds" >> $f
MMDX_STORE_WT=1 MMDX_SOAK_SEEDS=10000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -1 >> $f
echo "== 8000 rigs x 128 instances (device bone solve incl. CCD-IK vs the C oracle)" >> $f
timeout -k 10 900 python tools/soak_rig.py 8000 128 2>&1 | tail -1 >> $f
echo "== MMDX_SOLVE_DENSE=1, 2000 rigs x 128 instances" >> $f
MMDX_SOLVE_DENSE=1 timeout -k 10 600 python tools/soak_rig.py 2000 128 2>&1 | tail -1 >> $f
cat $f
