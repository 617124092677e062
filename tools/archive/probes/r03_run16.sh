#!/bin/bash
# Round 3, GPU call 26: SQ counters of the CCD-IK solver at 1 024 / 4 096 / 16 384 instances (VERDICT r02 task 7).
set -o pipefail
out=gpurun_out/r03/ik_sq; mkdir -p $out
export TMPDIR=/tmp
export RIG_ONLY=ik
for ni in 1024 4096 16384; do
  export RIG_NI=$ni
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/sq1_$ni -o sq1 -- python3 tools/rig_bench.py > $out/run_$ni.txt 2> $out/err_$ni.txt || { echo "pmc pass 1 failed at $ni"; tail -5 $out/err_$ni.txt; exit 1; }
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d $out/sq2_$ni -o sq2 -- python3 tools/rig_bench.py > /dev/null 2> $out/err2_$ni.txt || echo "pmc pass 2 failed at $ni (kept going)"
  echo "ni=$ni done"; ls $out/sq1_$ni | head -3
done
