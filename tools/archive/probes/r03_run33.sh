#!/bin/bash
# Round 3, GPU call 49: write-through flavour of the 512-thread crowd kernel (80 VGPRs: six waves per SIMD) with 8 / 12 / 16 / 4 instances per workgroup.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
AB_TRIES=1 AB_PLAIN_N=2 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  t512_g8=$V/libmmdx_wt512.so:FLAGS=32,MMDX_THREADS=512,MMDX_GROUP=8 t512_g12=$V/libmmdx_wt512b.so:FLAGS=32,MMDX_THREADS=512,MMDX_GROUP=12 \
  t512_g16=$V/libmmdx_wt512c.so:FLAGS=32,MMDX_THREADS=512,MMDX_GROUP=16 t512_g4=$V/libmmdx_wt512d.so:FLAGS=32,MMDX_THREADS=512,MMDX_GROUP=4 2>&1 | grep -v identical | tee $out/write_through_512_threads.txt
