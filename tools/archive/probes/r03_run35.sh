#!/bin/bash
# Round 3, GPU call 51: the whole GPU suite and the randomized soak with the write-through flavour forced wherever it exists (MMDX_STORE_WT=1).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
export MMDX_STORE_WT=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_crowd_store_policy_hints_and_defaults > $out/gputest_wt_forced.log 2>&1 || { tail -30 $out/gputest_wt_forced.log; exit 1; }
tail -2 $out/gputest_wt_forced.log
echo "== MMDX_STORE_WT=1: 3000 seeds of randomized models x every call form" > $out/soak_wt_forced.txt
MMDX_SOAK_SEEDS=3000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -2 >> $out/soak_wt_forced.txt
cat $out/soak_wt_forced.txt
