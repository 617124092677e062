#!/bin/bash
# Round 3, GPU call 41: tree = write-through + 8 instances per workgroup on arrays that are not fast.  A/B vs the hints; 32-byte vertex
# with small groups on such arrays; bench.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=c3 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  forced_nt=shipped:FLAGS=64 forced_wt=shipped:FLAGS=32 wt_g16=shipped:FLAGS=32,MMDX_GROUP=16 2>&1 | tee $out/adaptive_group_ab.txt
AB_TRIES=1 AB_PLAIN_N=3 AB_WORKLOAD=v32 AB_ROUNDS=5 AB_ITERS=30 AB_PLAIN=1 timeout -k 10 900 python tools/archive/probes/store_policy_ab.py \
  g8=shipped:MMDX_GROUP=8 g12=shipped:MMDX_GROUP=12 2>&1 | tee $out/v32_small_groups_ab.txt
timeout -k 10 400 python bench.py > $out/bench_k.json 2> $out/bench_k.err || { tail -20 $out/bench_k.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_k.json'))
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','cold_ms_per_step','plain_alloc_ms_per_step')})
print('frac',r['frac'],'step_frac',r['step_frac'],'placement',r['output_placement'],'plain',r.get('plain_alloc'))
PY
