#!/bin/bash
# Collects the evidence bench.py's roofline cites: run on the GPU box from the repo root.
#   bash tools/profile_round.sh rNN
# 1. plain bench line; 2. rocprofv3 --kernel-trace --stats of the same command; 3./4. separate PMC passes (FETCH_SIZE,
# WRITE_SIZE) as the microarchitecture guide prescribes; 5. SQ counter passes of the crowd step; 6. a trace with the
# secondary workloads on (all kernels of the library); 7. the per-instance-morph workloads alone (trace + PMC passes);
# 8. SQ counters of the rig kernels; 9. the single-frame kernels (trace + SQ counters).  Outputs in gpurun_out/<tag>/; tools/summarize_profiles.py condenses them.
# rocprofv3 is always given the program itself after "--" (python3 ...), never a wrapper.
set -e -o pipefail
tag=${1:-r03}
from=${2:-1}            # 1: everything; 6: from the all-workloads trace on (stages 1-5 already collected)
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export MMDX_BENCH_NO_GRAPH=1     # rocprofv3's tracer segfaults on hipStreamBeginCapture (ROCm 7.2): the graph leg of the extras is left out
if [ "$from" -le 1 ]; then
env -u MMDX_BENCH_NO_GRAPH python3 bench.py --steps 50 --warmup 5 > $out/bench.json
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B > $out/bench_under_rocprof.json 2> $out/rocprof.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc_sq1 -o sq1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq2 -o sq2 -- $B > /dev/null 2>&1
echo "crowd passes done" >&2
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_all -o all -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $out/rocprof_all.log
echo "all-workloads trace done" >&2
F="python3 tools/fused_bench.py c2 c5 c3p --iters 10"
$F > $out/fused_plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fused_kt -o kt -- $F > $out/fused_under_trace.txt 2> $out/fused_rocprof.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fused_fetch -o fetch -- $F > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/fused_write -o write -- $F > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/fused_sq1 -o sq1 -- $F > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/fused_sq2 -o sq2 -- $F > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/fused_tcc -o tcc -- $F > /dev/null 2>&1 || true
echo "fused passes done" >&2
R="python3 tools/rig_bench.py"
$R > $out/rig_plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rig_kt -o kt -- $R > /dev/null 2> $out/rig_rocprof.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/rig_sq1 -o sq1 -- $R > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d $out/rig_sq2 -o sq2 -- $R > /dev/null 2>&1 || true
echo "rig passes done" >&2
# 9. one frame of one model per launch (frame kernel for config 2, 512-thread tile kernel for config 5): trace + SQ counters
S="python3 tools/fused_bench.py c2x1 c5x1 --iters 50"
$S > $out/frame_plain.txt
MMDX_FRAME_KERNEL=0 MMDX_THREADS=256 $S > $out/frame_plain_round1_path.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/frame_kt -o kt -- $S > /dev/null 2> $out/frame_rocprof.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/frame_sq1 -o sq1 -- $S > /dev/null 2>&1 || true
echo "single-frame passes done" >&2
find $out -name "*.csv" | wc -l
