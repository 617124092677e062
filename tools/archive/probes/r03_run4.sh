#!/bin/bash
# Round 3, GPU call 4: GPU tests of the reactor pin / glue, placement counters (one process, many pairs), CCD-IK at larger crowds.
set -e -o pipefail
out=gpurun_out/r03
mkdir -p $out/placement_pmc2
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/gputest2.log 2>&1 || { tail -30 $out/gputest2.log; exit 1; }
tail -3 $out/gputest2.log
P="python3 tools/archive/probes/placement_counters.py"
n=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum"; do
  n=$((n+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/placement_pmc2/p$n -o p -- $P > $out/placement_pmc2/p$n.txt 2>&1 || { echo "pass $n ($set) failed"; tail -3 $out/placement_pmc2/p$n.txt; }
  echo "pass $n done"
done
python3 tools/archive/probes/placement_counters.py --analyze $out/placement_pmc2/p? > $out/placement_pmc2/summary.txt 2>&1 || true
cat $out/placement_pmc2/summary.txt
RIG_NI=1024,4096,16384 RIG_ONLY=ik timeout -k 10 300 python tools/rig_bench.py > $out/rig_scaling.txt 2>&1
cat $out/rig_scaling.txt
