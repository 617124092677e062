"""Condenses what tools/profile_round.sh collected (gpurun_out/<tag>/) into the small summaries
committed under profiles/<round>/:  python tools/summarize_profiles.py gpurun_out/r01b profiles/r01
  config3_kernel_stats.csv        rocprofv3 --kernel-trace --stats (verbatim)
  config3_pmc_hbm_traffic.csv     per-kernel mean/min/max of FETCH_SIZE and WRITE_SIZE (KB as reported,
                                  separate passes); bench.py reads this file for roofline.traffic
  all_workloads_kernel_stats.csv  the same with bench.py's secondary workloads on: every kernel of the library
  config3_bench.json              the plain bench line of the same box
  config3_bench_under_rocprof.json the line printed while tracing (HIP-event time to compare with the trace)
  config3_kernel_trace_timed_region.csv  per-kernel average over the `steps` dispatches of bench.py's timed region
                                  and over the uninstrumented repeat that follows it (the stats file above
                                  also averages the untimed settle / warm-up launches, which run through the
                                  clock transient)
"""
import collections
import csv
import json
import os
import shutil
import sys


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "config3_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "config3_bench.json"))
    if os.path.exists(os.path.join(src, "kt_all", "all_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "kt_all", "all_kernel_stats.csv"), os.path.join(dst, "all_workloads_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "config3_bench_under_rocprof.json"))
    line = json.load(open(os.path.join(src, "bench_under_rocprof.json")))
    steps, trailing = line["steps"], line["roofline"].get("trailing_steps", 0)
    per = collections.OrderedDict()
    with open(os.path.join(src, "kt", "kt_kernel_trace.csv"), newline="") as f:
        for r in csv.DictReader(f):
            per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(dst, "config3_kernel_trace_timed_region.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches_in_trace", "timed_n", "avg_ns_timed_region", "min_ns_timed_region",
                    "max_ns_timed_region", "avg_ns_uninstrumented_repeat", "avg_ns_all"])
        for k, v in per.items():
            if "deform_kernel" in k or "morph_apply" in k:
                t = v[len(v) - trailing - steps:len(v) - trailing]       # the K launches bracketed by events
                u = v[len(v) - trailing:len(v) - trailing + steps]      # the same K steps repeated without events
                w.writerow([k, len(v), len(t), sum(t) / len(t), min(t), max(t), sum(u) / len(u) if u else "", sum(v) / len(v)])
    rows = []
    for counter, path in (("FETCH_SIZE", "pmc_fetch/fetch_counter_collection.csv"),
                          ("WRITE_SIZE", "pmc_write/write_counter_collection.csv")):
        per = collections.OrderedDict()
        with open(os.path.join(src, path), newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            rows.append((counter, k, len(v), sum(v) / len(v), min(v), max(v)))
    with open(os.path.join(dst, "config3_pmc_hbm_traffic.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["counter", "kernel", "dispatches", "mean_KB", "min_KB", "max_KB"])
        w.writerows(rows)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
