#!/bin/bash
# Round 3, GPU call 25: final tree -- GPU suite, smoke, default bench, long soak (new seeds).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gputest_final4.log 2>&1 || { tail -30 $out/gputest_final.log; exit 1; }
tail -2 $out/gputest_final4.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > $out/bench_l.json 2> $out/bench_l.err || { tail -20 $out/bench_l.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_l.json'))
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','cold_ms_per_step','plain_alloc_ms_per_step')})
print('frac',r['frac'],'step_frac',r['step_frac'],'traffic',r['traffic'],'placement',r['output_placement'])
PY
echo "== soak: 1000 seeds of randomized models x every call form" > $out/soak_final4.txt
MMDX_SOAK_SEEDS=1000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -2 >> $out/soak_final4.txt
echo "== soak: 1000 rigs x 128 instances (device bone solve incl. CCD-IK vs the C oracle)" >> $out/soak_final4.txt
timeout -k 10 600 python tools/soak_rig.py 1000 128 2>&1 | tail -2 >> $out/soak_final4.txt
cat $out/soak_final4.txt
