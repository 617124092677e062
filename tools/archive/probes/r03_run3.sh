#!/bin/bash
# Round 3, GPU call 3: the slow / fast placement under the TLB (UTCL1/UTCL2) and L2 -> fabric write counters.
set -e -o pipefail
out=gpurun_out/r03/placement_pmc
mkdir -p $out
export TMPDIR=/tmp
P="python3 tools/archive/probes/placement_counters.py"
for mode in fast slow; do
$P $mode > $out/plain_$mode.txt 2>&1 || true
cat $out/plain_$mode.txt
n=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  n=$((n+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${mode}_$n -o p -- $P $mode > $out/${mode}_$n.txt 2>&1 || { echo "pass $n ($set) failed"; tail -3 $out/${mode}_$n.txt; }
  grep "chosen pair\|crowd kernel" $out/${mode}_$n.txt || true
done
done
find $out -name "*counter_collection.csv" | wc -l
