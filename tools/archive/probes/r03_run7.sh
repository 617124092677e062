#!/bin/bash
# Round 3, GPU call 7: instances per workgroup of the per-instance-morph kernels under nt stores.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
FB_SWEEP="MMDX_GROUP=8,16,24,32" FB_ROUNDS=5 timeout -k 10 400 python tools/fused_bench.py c2 c5 c3p --iters 10 > $out/fused_group_sweep.txt 2>&1
cat $out/fused_group_sweep.txt
FB_SWEEP="MMDX_LDS_TARGET=49152,65536,81920" FB_ROUNDS=5 timeout -k 10 400 python tools/fused_bench.py c2 c5 c3p --iters 10 > $out/fused_lds_sweep.txt 2>&1
cat $out/fused_lds_sweep.txt
