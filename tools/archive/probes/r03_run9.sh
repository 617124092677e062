#!/bin/bash
# Round 3, GPU call 9: soak of the shipped (nt) build -- 600 random models through every call form, 60 through both opt-in modes,
# 1 000 random rigs x 128 instances -- and a default bench line (counter-side traffic of the secondary workloads now attached).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
MMDX_SOAK_SEEDS=600 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "randomized_models_all_call_forms" > $out/soak_models.txt 2>&1; echo "models soak rc=$?"; tail -2 $out/soak_models.txt
MMDX_SOAK_SEEDS=60 timeout -k 10 600 python -m pytest tests/test_tile_order.py -m gpu -q -k "randomized" > $out/soak_modes.txt 2>&1; echo "modes soak rc=$?"; tail -2 $out/soak_modes.txt
timeout -k 10 900 python tools/soak_rig.py 1000 128 > $out/soak_rig.txt 2>&1; echo "rig soak rc=$?"; tail -4 $out/soak_rig.txt
timeout -k 10 500 python bench.py > $out/bench_c.json 2> $out/bench_c.err || tail -5 $out/bench_c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_c.json'))
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','cold_ms_per_step','plain_alloc_ms_per_step')})
print({k:r[k] for k in ('frac','step_frac','avg_kernel_ms','traffic','output_placement')})
o=d['other_workloads']
for k in ('config2_64_frames_per_launch','config3prime_per_instance_morphs','config5_64_frames_per_launch_fp16'):
    print(k,{x:o[k].get(x) for x in ('ms_per_call','frac_of_8TBs','pmc_GBs','pmc_frac_of_8TBs','pmc_traffic_over_algorithmic','pmc_traffic_withheld')})
PY
