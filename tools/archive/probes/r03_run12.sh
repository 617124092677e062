#!/bin/bash
# Round 3, GPU call 21: 1024-vertex tiles (build variant) for the crowd kernel, 512 threads x 2 slots per lane.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
V=build/variants
for wl in c3 v32; do
AB_WORKLOAD=$wl AB_ROUNDS=7 AB_ITERS=30 AB_PLAIN=0 timeout -k 10 300 python tools/archive/probes/store_policy_ab.py \
  t1024_512x16=$V/libmmdx_t1024.so:MMDX_THREADS=512,MMDX_GROUP=16 \
  t1024_512x8=$V/libmmdx_t1024b.so:MMDX_THREADS=512,MMDX_GROUP=8 \
  t1024_512x24=$V/libmmdx_t1024c.so:MMDX_THREADS=512,MMDX_GROUP=24 \
  t1024_256x16=$V/libmmdx_t1024d.so:MMDX_GROUP=16 2>&1 | tee -a $out/tile1024_ab.txt
done
