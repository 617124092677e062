#!/bin/bash
# Round 3, GPU call 5: per-channel L2 counters of fast vs slow placements, the graph-under-profiler diagnostic (one run), bench line.
set -o pipefail
out=gpurun_out/r03
mkdir -p $out/placement_pmc3
export TMPDIR=/tmp
P="python3 tools/archive/probes/placement_counters.py"
timeout -k 10 150 rocprofv3 --pmc TCC_EA0_WRREQ TCC_TAG_STALL TCC_BUSY TCC_EA0_WRREQ_STALL --kernel-trace --output-format csv json -d $out/placement_pmc3/raw -o p -- $P > $out/placement_pmc3/raw.txt 2>&1 || { echo "raw pass failed"; tail -3 $out/placement_pmc3/raw.txt; }
ls -la $out/placement_pmc3/raw/
timeout -k 10 500 python bench.py > $out/bench_b.json 2> $out/bench_b.err || tail -5 $out/bench_b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_b.json'))
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','cold_ms_per_step','plain_alloc_ms_per_step')})
print({k:r[k] for k in ('frac','step_frac','avg_kernel_ms','measured_store_pattern_GBs','measured_fill_GBs','cold','output_placement')})
PY
mkdir -p $out/graph_diag
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $out/graph_diag -o g -- python3 -X faulthandler tools/archive/probes/graph_under_profiler.py > $out/graph_diag/stdout.txt 2> $out/graph_diag/stderr.txt
echo "graph diagnostic exit code $?" | tee $out/graph_diag/exit.txt
grep MARK $out/graph_diag/stdout.txt | tail -3
