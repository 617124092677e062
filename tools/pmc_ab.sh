#!/bin/bash
# SQ / LDS / L2 counters of a fused_bench workload under several environments, one rocprofv3 --pmc pass per counter group and
# variant (counters only with --kernel-trace, as the pool requires); tools/pmc_summary.py condenses the result.
#   bash tools/pmc_ab.sh <tag> <workloads | rig> "NAME=ENV=VAL ENV2=VAL2" ["NAME2=..."]
set -e -o pipefail
tag=$1; shift
wl=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  [ "$envs" = "-" ] && envs=""
  B="python3 tools/fused_bench.py $wl --iters 6"
  [ "$wl" = "rig" ] && B="python3 tools/rig_bench.py"        # (RIG_ONLY / RIG_NI from the environment)
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ( export $envs; rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_${name}_$i -o p -- $B > $out/pmc_${name}_$i.txt 2>&1 ) || echo "pass $i of $name failed"
  done
done
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt
cat $out/pmc_summary.txt
