#!/bin/bash
# SQ counters of the crowd step's kernels: bash tools/pmc_crowd.sh <tag> [ENV=VALUE ...] -> gpurun_out/<tag>/pmc_crowd_*
set -e -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-settle --plain-alloc"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LEVEL_WAVES --kernel-trace --output-format csv -d $out/pmc_crowd_sq1 -o sq1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_crowd_sq2 -o sq2 -- $B > /dev/null 2>&1
python3 - $out <<'PY'
import csv, collections, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_crowd_sq*/*counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "deform_kernel" not in k and "crowd_kernel" not in k: continue
        k = k.split("::")[-1].split("(")[0]
        per[(k, r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        dur = None
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
