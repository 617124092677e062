#!/usr/bin/env python3
"""Per kernel and variant: mean of every counter tools/pmc_ab.sh collected (gpurun_out/<tag>/pmc_<name>_<pass>/).
    python tools/pmc_summary.py gpurun_out/<tag>"""
import collections
import csv
import glob
import os
import re
import sys


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))      # (variant, kernel, grid) -> counter -> values
    for d in sorted(glob.glob(os.path.join(root, "pmc_*_*"))):
        if not os.path.isdir(d):
            continue
        variant = re.sub(r"_\d+$", "", os.path.basename(d)[4:])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                k = r["Kernel_Name"]
                if not any(w in k for w in ("deform_kernel", "pack_kernel", "skeleton_ordered_kernel", "ik_coop_kernel")):
                    continue
                k = re.sub(r"^.*?((deform|pack|skeleton_ordered)_kernel<[^>]*>|ik_coop_kernel).*$", r"\1", k)
                acc[(variant, k, r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (variant, k, grid), ctr in sorted(acc.items()):
        print(f"== {variant}  {k}  grid {grid}")
        for c, v in sorted(ctr.items()):
            print(f"   {c:24s} {sum(v) / len(v):16.1f}   (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1])
