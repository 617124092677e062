#!/bin/bash
# rocprofv3 evidence for the per-instance-morph ("fused gather") deform kernels: kernel trace + separate PMC passes.
#   bash tools/profile_fused.sh <tag>      -> gpurun_out/<tag>/fused_*
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 tools/fused_bench.py c2 c5 c3p --iters 10"
$B > $out/fused_plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fused_kt -o kt -- $B > $out/fused_under_trace.txt 2> $out/fused_rocprof.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fused_fetch -o fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/fused_write -o write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $out/fused_sq1 -o sq1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/fused_sq2 -o sq2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/fused_tcc -o tcc -- $B > /dev/null 2>&1 || true
find $out -name "*.csv" | sort
