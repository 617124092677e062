#!/bin/bash
# Round 3, GPU call 35: adaptive store flavour (nt on fast arrays, sc1 nt otherwise) in the tree: new test, suite, A/B vs the hints, bench.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gputest_wt.log 2>&1 || { tail -30 $out/gputest_wt.log; exit 1; }
tail -2 $out/gputest_wt.log
AB_TRIES=64 AB_WORKLOAD=c3 AB_ROUNDS=9 AB_ITERS=40 AB_PLAIN=1 timeout -k 10 400 python tools/archive/probes/store_policy_ab.py forced_nt=shipped:FLAGS=64 forced_wt=shipped:FLAGS=32 2>&1 | tee $out/store_policy_adaptive_ab.txt
timeout -k 10 400 python bench.py > $out/bench_i.json 2> $out/bench_i.err || { tail -20 $out/bench_i.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_i.json'))
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','cold_ms_per_step','plain_alloc_ms_per_step')})
print('frac',r['frac'],'step_frac',r['step_frac'],'traffic',r['traffic'],'placement',r['output_placement'],'plain',r.get('plain_alloc'))
PY
