#!/bin/bash
# Collects the evidence bench.py's roofline cites: run on the GPU box from the repo root.
#   bash tools/profile_round.sh rNN
# 1. plain bench line; 2. rocprofv3 --kernel-trace --stats of the same command; 3./4. separate PMC
# passes (FETCH_SIZE, WRITE_SIZE) as the microarchitecture guide prescribes; 5. a trace with the secondary
# workloads on (all kernels of the library).  Outputs in gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 50 --warmup 5 > $out/bench.json
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B > $out/bench_under_rocprof.json 2> $out/rocprof.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o write -- $B > /dev/null 2>&1
# 5. every kernel of the library in one trace: the same run with the secondary workloads (bench extras) on
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_all -o all -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $out/rocprof_all.log
find $out -name "*.csv" | sort
