"""Condenses what tools/profile_round.sh collected (gpurun_out/<tag>/) into the small summaries
committed under profiles/<round>/:  python tools/summarize_profiles.py gpurun_out/r01b profiles/r01
  config3_kernel_stats.csv        rocprofv3 --kernel-trace --stats (verbatim)
  config3_pmc_hbm_traffic.csv     per-kernel mean/min/max of FETCH_SIZE and WRITE_SIZE (KB as reported,
                                  separate passes); bench.py reads this file for roofline.traffic
  all_workloads_kernel_stats.csv  the same with bench.py's secondary workloads on: every kernel of the library
  config3_bench.json              the plain bench line of the same box
  config3_bench_under_rocprof.json the line printed while tracing (HIP-event time to compare with the trace)
  config3_kernel_trace_timed_region.csv  deform-kernel average per segment of bench.py's run (timed region, the
                                  kernel-only launches behind roofline.frac, the event-bracketed repeat; the
                                  stats file above also averages the untimed settle / warm-up launches, which
                                  run through the clock transient)
  config3_pmc_hbm_traffic.meta.json  build + workload the PMC passes were collected for (bench.py withholds
                                  roofline.traffic when they do not match its own run)
"""
import collections
import re
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
    segs = line["roofline"]["trace_segments"]          # deform-kernel launches at the end of the run, oldest first
    # every launch of the deform kernel, whichever store flavour it was instantiated with (cached / write-through: the bench's
    # plain-allocation leg usually runs the other one), in launch order; the morph pass separately
    deform, morph = [], []
    with open(os.path.join(src, "kt", "kt_kernel_trace.csv"), newline="") as f:
        for r in csv.DictReader(f):
            t = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
            if "deform_kernel" in t[2]:
                deform.append(t)
            elif "morph_apply" in t[2]:
                morph.append(t)
    deform.sort()
    with open(os.path.join(dst, "config3_kernel_trace_timed_region.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches_in_trace", "segment", "n", "avg_ns", "min_ns", "max_ns", "avg_ns_all"])
        pos = len(deform) - sum(n for _, n in segs)
        all_avg = sum(d for _, d, _ in deform) / len(deform)
        for name, n in segs:
            part = deform[pos:pos + n]
            pos += n
            t = [d for _, d, _ in part]
            kernels = sorted({k for _, _, k in part})
            w.writerow([" | ".join(kernels), len(deform), name, len(t), sum(t) / len(t), min(t), max(t), all_avg])
        if morph:
            v = [d for _, d, _ in morph]
            w.writerow([morph[0][2], len(v), "all", len(v), sum(v) / len(v), min(v), max(v), sum(v) / len(v)])
    # provenance of the PMC figures bench.py quotes as roofline.traffic: the build and workload they were collected for
    with open(os.path.join(dst, "config3_pmc_hbm_traffic.meta.json"), "w") as f:
        json.dump({"kernel_source_sha": line["roofline"]["kernel_source_sha"],
                   "instances_per_gpu": line["config"]["instances_per_gpu"], "vertices": line["config"]["vertices"],
                   "collected_with": "tools/profile_round.sh (separate --pmc passes)"}, f, indent=1)
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


def counter_summary(src, passes, dst_csv, want):
    """Mean of every counter per kernel (name shortened) over the dispatches of separate --pmc passes."""
    rows = []
    for sub in passes:
        path = [os.path.join(src, sub, f) for f in os.listdir(os.path.join(src, sub))] if os.path.isdir(os.path.join(src, sub)) else []
        path = [q for q in path if q.endswith("counter_collection.csv")]
        if not path:
            continue
        per = collections.OrderedDict()
        with open(path[0], newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if not any(w in k for w in want):
                    continue
                mm = re.search(r"(\w+_kernel(?:<[^>]*>)?)", k)
                key = (mm.group(1) if mm else k, r["Grid_Size"], r["Workgroup_Size"])
                per.setdefault(key, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for (k, grid, wg), cs in per.items():
            for c, v in cs.items():
                rows.append((k, grid, wg, c, len(v), sum(v) / len(v), min(v), max(v)))
    with open(dst_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "workgroup", "counter", "dispatches", "mean", "min", "max"])
        w.writerows(rows)


def extras(src, dst):
    """Round 2 additions: SQ counters of the crowd step, the per-instance-morph kernels (trace + PMC), the rig kernels."""
    counter_summary(src, ["pmc_sq1", "pmc_sq2"], os.path.join(dst, "config3_pmc_sq.csv"), ["deform_kernel", "morph_apply"])
    counter_summary(src, ["fused_fetch", "fused_write", "fused_sq1", "fused_sq2", "fused_tcc"],
                    os.path.join(dst, "fused_gather_pmc.csv"), ["deform_kernel", "flatten"])
    # provenance for bench.py's counter-side figures of the per-instance-morph workloads (fused_pmc_traffic)
    line = json.load(open(os.path.join(src, "bench_under_rocprof.json")))
    with open(os.path.join(dst, "fused_gather_pmc.meta.json"), "w") as f:
        json.dump({"kernel_source_sha": line["roofline"]["kernel_source_sha"],
                   "collected_with": "tools/profile_round.sh stage 7 (tools/fused_bench.py c2 c5 c3p, separate --pmc passes)"}, f, indent=1)
    counter_summary(src, ["rig_sq1", "rig_sq2"], os.path.join(dst, "rig_pmc_sq.csv"), ["skeleton", "bone_track"])
    counter_summary(src, ["frame_sq1"], os.path.join(dst, "single_frame_pmc_sq.csv"), ["frame_kernel", "deform_kernel"])
    for a, b in (("fused_kt/kt_kernel_stats.csv", "fused_gather_kernel_stats.csv"), ("fused_plain.txt", "fused_gather_bench.txt"),
                 ("rig_kt/kt_kernel_stats.csv", "rig_kernel_stats.csv"), ("rig_plain.txt", "rig_bench.txt"),
                 ("frame_kt/kt_kernel_stats.csv", "single_frame_kernel_stats.csv"), ("frame_plain.txt", "single_frame_bench.txt"),
                 ("frame_plain_round1_path.txt", "single_frame_bench_round1_path.txt")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
    extras(sys.argv[1], sys.argv[2])
