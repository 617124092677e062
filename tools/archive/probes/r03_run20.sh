#!/bin/bash
# Round 3, GPU call 32: the rig stage of tools/profile_round.sh again (the solver gained a variant and a smaller LDS budget).
set -o pipefail
out=gpurun_out/r03t; mkdir -p $out
export TMPDIR=/tmp
R="python3 tools/rig_bench.py"
timeout -k 10 200 $R > $out/rig_plain.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rig_kt -o kt -- $R > /dev/null 2> $out/rig_rocprof.log || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/rig_sq1 -o sq1 -- $R > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d $out/rig_sq2 -o sq2 -- $R > /dev/null 2>&1 || true
cat $out/rig_plain.txt
