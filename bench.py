#!/usr/bin/env python3
"""bench.py -- skinned vertices/sec of the deformation hot path on MI355X, with the HBM roofline of
the dominant kernel and the reference CPU path timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3], SURVEY.md section 8d "config 3/4"): a crowd of 1024 instances PER GPU
of the synthetic 50 000-vertex / 300-bone / 200-morph model, per-instance bone palettes resident in
HBM, one shared morph state, outputs = Poser::pose_image (f32 positions + normals, 24 B / vertex)
resident in HBM.  One STEP = one mmdx_deform_batched() call = the whole hot path over the crowd
(group-morph flatten -> shared morph pass -> skinning + write-out).  Instances shard across ranks
with no data-path collective ("scaling": "weak"); torch.distributed (gloo) is used only for the
start/stop barriers and the max-over-ranks reduction of the elapsed time.

The config-2 (single 50k model, latency) and config-5 (256k verts, fp16 positions) figures ride
along under "other_workloads"; they are not the headline value.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s measured copy


def algorithmic_bytes_config3(nv, nb, nm, ne, ni, n1, n2, n4):
    """SURVEY.md section 8d, minimal encodings.  Returns (deform kernel bytes, whole step bytes)."""
    static = nv * (12 + 12 + 1) + n1 * 2 + n2 * 8 + n4 * 24
    deform = static + nv * 12 + ni * (nv * 24 + nb * 48)          # + shared morphed positions read
    morph = ne * 16 + nv * 12 + nm * 4                            # table + morphed write + weights
    return deform, deform + morph


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: THIS process never loads the product library or touches
    the GPU; it starts N fresh rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would hand them), relays rank 0's JSON line and returns the
    first non-zero exit code.  Never a re-exec of a process that has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    # Every rank gets its own contiguous slice of the host's CPUs (on the 8-GPU nodes consecutive GPUs hang off the same
    # socket, and the CPU list is socket-major): a rank's launch thread then neither migrates nor shares a core with
    # another rank's.  MMDX_BENCH_NO_AFFINITY=1 leaves the scheduler alone.
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // n if os.environ.get("MMDX_BENCH_NO_AFFINITY") != "1" else 0
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", "1")
        mine = set(cpus[r * per:(r + 1) * per]) if per >= 1 else None
        if mine:
            env["MMDX_BENCH_PINNED"] = "1"          # (the rank must not slice its slice again)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      preexec_fn=(lambda m=mine: os.sched_setaffinity(0, m)) if mine else None))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    bad = [c for c in codes if c != 0]
    if bad:
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return bad[0] if bad[0] > 0 else 1
    return 0 if lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--instances-per-gpu", type=int, default=1024)
    ap.add_argument("--layout", choices=["soa", "vertex32"], default="soa")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--plain-alloc", action="store_true", help="plain hipMalloc for the output arrays")
    ap.add_argument("--shop-alloc", type=int, default=128, metavar="TRIES",
                    help="let mmdx_crowd_output_alloc try up to TRIES placements of the output arrays (bounded: ~5 ms per try, "
                         "every rank on its own GPU; the same bound at every N, so that per-N values compare like with like)")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed settle batches before the warm-up")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous, sharding and the JSON line only -- no GPU work (CPU test of the N>1 plumbing)")
    args = ap.parse_args()

    # N > 1 without a launcher: decided before the product library is loaded or any HIP call is made.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # Under an external launcher (torch.distributed.run) nobody pinned the ranks: take this rank's slice of the CPUs the job
    # was given, as launch_ranks does for its own children (no-op for one rank or when the slice would be empty).
    try:
        lw, lr = int(os.environ.get("LOCAL_WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
        cpus = sorted(os.sched_getaffinity(0))
        if (lw > 1 and os.environ.get("MMDX_BENCH_NO_AFFINITY") != "1" and os.environ.get("MMDX_BENCH_PINNED") != "1"
                and len(cpus) >= 2 * lw):
            per = len(cpus) // lw
            os.sched_setaffinity(0, set(cpus[lr * per:(lr + 1) * per]))
    except (OSError, ValueError):
        pass

    from simple_mmd_renderer_amd.crowd import Rendezvous, crowd_frames, shard_instances
    if args.dry_run:
        rv = Rendezvous()
        lo, hi = shard_instances(args.instances_per_gpu * rv.world, rv.world, rv.rank)
        rv.barrier()
        covered = rv.sum(hi - lo)
        ranges = rv.gather_u64([lo, hi])
        rv.close()
        if rv.rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": rv.world, "instances": int(covered), "ranges": ranges,
                              "steps": args.steps, "warmup": args.warmup, "scaling": "weak"}))
        return

    # The product library first (it binds the HIP runtime at load); torch only for rendezvous.
    from simple_mmd_renderer_amd import _capi as api
    from simple_mmd_renderer_amd import build, synth
    from simple_mmd_renderer_amd.engine import (DeformModel, DeviceBuffer, device_count, device_name,
                                                device_select, device_synchronize)
    build.build()
    api.lib()
    stamp = api.check_library_matches_tree()        # the loaded library was built from THIS tree's sources, or we stop here
    # HIP first, torch second: torch bundles its own HIP runtime; if it gets to initialise before this
    # process has touched the GPU through the system runtime libmmdx links against, the latter sees no
    # device.  So count / select / touch the device, THEN import torch for the gloo rendezvous.
    ndev = device_count()
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible -- this engine has no CPU path to fall back to")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device_select(local_rank % ndev)
    DeviceBuffer(256).free()
    rv = Rendezvous()
    rank, world = rv.rank, rv.world
    args.gpus = world           # under a launcher the world size is the launcher's
    barrier = rv.barrier

    # ---- workload ------------------------------------------------------------------------------
    model = synth.make_config("config3_crowd")
    ni = args.instances_per_gpu
    lo, hi = shard_instances(ni * world, world, rank)
    pals = synth.make_palettes(model, crowd_frames(lo, hi))   # phase-shifted animation per instance
    rates = synth.morph_weights(model.nm, 30)[0]
    layout = api.OUT_SOA if args.layout == "soa" else api.OUT_VERTEX32
    pos_scale = 1.0 if layout == api.OUT_SOA else 0.1

    dm = DeformModel(model)
    info = dm.info
    # Output arrays through the engine's placement-aware allocator: on MI355X the store rate of the crowd
    # pattern depends on where the driver puts the arrays (bimodal); set-up work,
    # outside the timed region, and done first: the big arrays of a young process land in the fast mode
    # within a try or two (tools/archive/probes/shop_probe.py).  --plain-alloc takes whatever hipMalloc hands out first.
    d_a, d_b, placement = dm.alloc_outputs(layout, ni, 1 if args.plain_alloc else args.shop_alloc)
    d_pal = DeviceBuffer.from_numpy(pals)
    d_w = DeviceBuffer.from_numpy(rates)
    # the probe's verdict on these arrays travels with the call (mmdx.h: the library keeps no table of addresses)
    base_flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    flags = base_flags | placement.get("store_flags", 0)
    rates_b = rates.copy()
    rates_b[0] = np.float32(0.25) if rates_b[0] != np.float32(0.25) else np.float32(0.5)
    d_w2 = DeviceBuffer.from_numpy(rates_b)
    flip = [0]

    def step():
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr if d_b else None, layout, flags,
                              pos_scale)

    def kernel_only_step():                 # the deform kernel alone: the morphed positions of the last step are reused
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr if d_b else None, layout,
                              flags | api.MORPH_UNCHANGED, pos_scale)

    def changing_step():                    # the shared morph state differs from the previous step's: the morph pass walks every time
        flip[0] ^= 1
        dm.deform_batched_raw(ni, (d_w2 if flip[0] else d_w).ptr, d_pal.ptr, d_a.ptr, d_b.ptr if d_b else None, layout, flags,
                              pos_scale)

    def timed_batch(fn, n):
        """n back-to-back calls bracketed by ONE pair of HIP events on the launch stream; ms per call."""
        dm.timer_start()
        for _ in range(n):
            fn()
        return dm.timer_stop() / n

    # Settle: the first ~100 launches after an idle period run through a clock / power transient (step time
    # overshoots by 10-25 % around launch 10-30 and decays; tools/archive/probes/alloc_kernel_probe.py), so a 50-step
    # measurement taken cold reports the transient, not the sustained rate.  Untimed batches of 20 steps until two
    # consecutive batches agree within 1.5 % and stop improving (at most 600 steps, ~0.15 s), then the contract's
    # W warm-up steps and the K timed steps.
    # ---- cold figure first: the contract's W warm-up steps and K timed steps taken right after set-up, BEFORE any settle
    # batch -- what a caller gets who starts stepping at once (reported as cold_ms_per_step next to the headline) ------------
    for _ in range(args.warmup):
        step()
    device_synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_synchronize()
    cold_ms = rv.max(time.perf_counter() - t0) / args.steps * 1e3
    settle_batches = []
    if not args.no_settle:
        for _ in range(30):
            settle_batches.append(timed_batch(step, 20))
            # settled = two consecutive batches within 1.5 % of their predecessors AND no longer improving (the
            # transient decays monotonically; a new batch that still beats every earlier one by 0.5 % is its tail)
            if (len(settle_batches) >= 3 and all(abs(settle_batches[-k] / settle_batches[-k - 1] - 1) < 0.015 for k in (1, 2))
                    and settle_batches[-1] > 0.995 * min(settle_batches[:-1])):
                break
    # ranks settle in different numbers of batches: whoever is done keeps stepping until all are, so that no GPU
    # idles (and drops its clocks) at the barrier in front of the timed region
    def keep_busy():
        for _ in range(5):
            step()
        device_synchronize()
    rv.barrier_while(keep_busy)
    for _ in range(args.warmup):
        step()
    device_synchronize()
    # ---- the timed region: K whole steps, back to back, nothing else on the stream -----------------------------
    # One HIP event before the first launch and one after the last (on the launch stream) plus the host's wall clock
    # around barrier + synchronize.  No per-kernel events in here: round 1 showed that each event record opens a
    # ~10 us bubble on the stream in which the previous kernel's stores drain, which makes the step ~5 % longer and
    # the kernel inside it ~4 % SHORTER than in the stream a user runs (profiles/r01/event_overhead_trace.txt).
    barrier()
    t0 = time.perf_counter()
    dm.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = dm.timer_stop()                 # HIP events on the launch stream; also drains it
    device_synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = rv.max(elapsed)
    # ---- the dominant kernel, live: K back-to-back launches of the deform kernel ALONE between two HIP events on
    # its stream (MMDX_MORPH_UNCHANGED: the shared morph state did not change, the morph pass is skipped) -- the
    # kernel's duration in an uninstrumented stream, which is what a rocprofv3 kernel trace of this run shows too.
    # Three batches of K, the median is reported (one batch now and then catches a clock or placement hiccup: 0.2275 vs 0.2185-0.2198 ms
    # on otherwise identical runs); all three are kept in roofline.kernel_only_batches_ms.
    pass_stats = dm.morph_pass_stats()      # (walks, device-side skips, host-side skips) up to the end of the timed region
    kernel_batches = [timed_batch(kernel_only_step, args.steps) for _ in range(3)]
    kernel_ms = sorted(kernel_batches)[1]
    # ---- the same K steps with a morph state that CHANGES every step (two rate sets taking turns): flatten + walk + deform -----
    timed_batch(changing_step, 5)
    changing_ms = sorted(timed_batch(changing_step, args.steps) for _ in range(3))[1]
    step()                                  # leave the headline's morph state behind
    # ---- the old figure, kept as a named extra: events around every kernel of K more steps ---------------------
    dm.profile_enable(True)
    for _ in range(args.steps):
        step()
    ncalls, skin_total, morph_total = dm.profile_collect()
    dm.profile_enable(False)
    skin_avg = skin_total / ncalls
    morph_avg = morph_total / ncalls

    # cached non-temporal stores into arrays the placement probe found fast, write-through stores otherwise (mmdx.h,
    # MMDX_OUT_STORES_*): what the library chose for these arrays
    placement = dict(placement, store_policy=dm.last_store_policy())
    total_vertices = float(ni) * world * model.nv * args.steps
    value = total_vertices / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    deform_bytes, walk_step_bytes = algorithmic_bytes_config3(model.nv, model.nb, model.nm, info.n_entries, ni,
                                                              info.n_bdef1, info.n_bdef2, info.n_bdef4)
    # Config 3 as BASELINE states it: ONE shared morph state, per-instance palettes.  The state does not change across the timed
    # steps, the library notices (device-side comparison of the rates, mmdx.h MMDX_MORPH_UNCHANGED) and the morph pass skips its
    # walk: the step's compulsory bytes are the deform kernel's plus the rates read twice.  The walking form is timed beside it.
    walked_every_step = pass_stats[1] == 0 and pass_stats[2] == 0
    step_bytes = walk_step_bytes if walked_every_step else deform_bytes + 2 * model.nm * 4
    if layout == api.OUT_VERTEX32:
        deform_bytes += ni * model.nv * 8 + model.nv * 8        # 32 B out + uv in
        step_bytes += ni * model.nv * 8 + model.nv * 8
    achieved = deform_bytes / (kernel_ms * 1e-3) / 1e9

    result = {
        "metric": "skinned vertices/sec (instance-sharded crowd); achieved HBM GB/s vs roofline",
        "value": value, "unit": "vertices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "cold_ms_per_step": cold_ms, "plain_alloc_ms_per_step": None,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config3: 1024-instance crowd per GPU of the 50k-vert/300-bone/200-morph "
                               "model, shared morph state, per-instance palettes in HBM",
                   "instances_per_gpu": ni, "vertices": model.nv, "bones": model.nb, "morphs": model.nm,
                   "morph_entries": int(info.n_entries), "out_layout": args.layout,
                   "sharding": f"instances/{world} ranks, no collective"},
        "roofline": {"bound": "hbm", "kernel": "deform_kernel (skinning + write-out)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": deform_bytes, "avg_kernel_ms": kernel_ms,
                     "avg_kernel_ms_how": f"median of 3 batches of {args.steps} back-to-back launches of the kernel alone, each batch "
                                          "between two HIP events on its stream (rank 0)",
                     "kernel_only_batches_ms": [round(x, 5) for x in kernel_batches],
                     # the whole step (morph pass + deform kernel) against the same peak, from the driver-checkable
                     # ms_per_step: kernel and step figures must tell one story
                     "step_algorithmic_bytes": step_bytes,
                     "step_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "step_event_ms": ev_ms / args.steps,
                     "step_form": ("default call form (device-resident shared rates, no MMDX_MORPH_UNCHANGED): " +
                                   ("the morph pass walked the table on every step" if walked_every_step else
                                    "rates unchanged across the timed steps, detected by the library on the device -- the morph "
                                    "pass launched every step and skipped its walk")),
                     "morph_pass_stats_at_end_of_timed_region": {"walks": pass_stats[0], "device_skips": pass_stats[1],
                                                                 "host_skips": pass_stats[2]},
                     # the morph state changes on every step (two rate sets in turn): flatten + walk + deform, median of 3 x K
                     "changing_morph_state": {"ms_per_step": changing_ms, "step_algorithmic_bytes": walk_step_bytes,
                                              "step_frac": walk_step_bytes / (changing_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     # W warm-up + K steps right after set-up, no settle batches in front (max over ranks, wall clock)
                     "cold": {"ms_per_step": cold_ms, "step_frac": step_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     # per-kernel events on every launch (round 1's headline; flatters the kernel, see above)
                     "event_bracketed_kernel_ms": skin_avg, "event_bracketed_morph_pass_ms": morph_avg,
                     "event_bracketed_frac": deform_bytes / (skin_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     # deform-kernel launches after the timed region, oldest first (trace post-processing)
                     "trace_segments": [["timed", args.steps], ["kernel_only", args.steps], ["kernel_only", args.steps],
                                        ["kernel_only", args.steps], ["changing_morph_state", 5 + 3 * args.steps + 1],
                                        ["event_bracketed", args.steps]],
                     "output_placement": placement,
                     "kernel_source_sha": kernel_source_sha(),
                     # the binary's own word on what it was built from, next to the hash of the tree (bench refuses a mismatch)
                     "library_source_sha": stamp["library_source_sha"], "tree_source_sha": stamp["tree_source_sha"],
                     "settle_batches_step_ms": [round(x, 4) for x in settle_batches]},
    }

    if world > 1:       # every rank placed its own output arrays: how many tries each took and what it got
        tries = rv.gather_u64([int(placement.get("tries", 0)), int(placement.get("store_GBs", 0.0))])
        if rank == 0:
            result["roofline"]["output_placement_per_rank"] = tries
    if rank == 0:
        result["roofline"].update(pmc_traffic(layout == api.OUT_SOA, ni, model.nv, placement.get("store_policy", "nt")))
        result["device"] = device_name(local_rank % ndev)
        # practical ceilings on this box (SURVEY.md section 8d asks for them next to the spec peak)
        nb_ceiling = 1 << 30
        src, dst = DeviceBuffer(nb_ceiling), DeviceBuffer(nb_ceiling)
        src.memset(1)
        import ctypes as C
        ms = C.c_float(0)
        api.check(api.lib().mmdx_bench_copy(dst.ptr, src.ptr, nb_ceiling, 10, C.byref(ms)))
        result["roofline"]["measured_copy_GBs"] = 2 * nb_ceiling / (ms.value * 1e-3) / 1e9
        api.check(api.lib().mmdx_bench_fill(dst.ptr, nb_ceiling, 10, C.byref(ms)))
        result["roofline"]["measured_fill_GBs"] = nb_ceiling / (ms.value * 1e-3) / 1e9
        src.free()
        dst.free()
        if layout == api.OUT_SOA and model.nv % 4 == 0:
            api.check(api.lib().mmdx_bench_store_pattern(d_a.ptr, d_b.ptr, model.nv, ni, 10, C.byref(ms)))
            result["roofline"]["measured_store_pattern_GBs"] = ni * model.nv * 24 / (ms.value * 1e-3) / 1e9
            step()      # leave real results in the output buffers
            result["roofline"]["trace_segments"].append(["trailing", 1])
        # What a caller sees who simply hipMallocs the output arrays (first placement, no probing): the same K steps
        # into plainly allocated arrays, next to the headline's placement-probed ones.  On MI355X about six placements
        # in seven store ~25 % slower for this two-array pattern; the headline says how many
        # tries its placement took.
        if not args.plain_alloc and placement.get("tries", 1) >= 1:
            p_a, p_b, p_info = dm.alloc_outputs(layout, ni, 1)

            def plain_step():
                # (no hint: these arrays were not probed, the library's default for unknown arrays applies)
                dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, p_a.ptr, p_b.ptr if p_b else None, layout,
                                      base_flags | p_info.get("store_flags", 0), pos_scale)
            timed_batch(plain_step, 20)
            pl_ms = timed_batch(plain_step, args.steps)
            pl = {"ms_per_step": pl_ms, "vertices_per_s": ni * model.nv / (pl_ms * 1e-3),
                  "step_frac": step_bytes / (pl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "store_policy": dm.last_store_policy()}
            if layout == api.OUT_SOA and model.nv % 4 == 0:
                api.check(api.lib().mmdx_bench_store_pattern(p_a.ptr, p_b.ptr, model.nv, ni, 10, C.byref(ms)))
                pl["store_pattern_GBs"] = ni * model.nv * 24 / (ms.value * 1e-3) / 1e9
            result["roofline"]["plain_alloc"] = pl
            result["plain_alloc_ms_per_step"] = pl_ms
            result["roofline"]["trace_segments"].append(["plain_alloc", 20 + args.steps])
            p_a.free()
            if p_b:
                p_b.free()

    # ---- CPU baseline: rank 0, N=1 only ------------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.pyoracle import Oracle, Reference, reference_available   # checker, CPU leg only
        reps = 24                                   # ~11 s of single-thread CPU work
        if reference_available():
            ref = Reference(model, normalize=True)
            secs = sum(ref.time_crowd(rates, pals) for _ in range(reps))
            ref.close()
            kind = "reference"
            what = "libmmd Poser (g++ -O2): 1 morph pass + 1024 x {palette inject, Deform()}"
        else:
            orc = Oracle()
            secs = sum(orc.time_crowd(model, rates, pals) for _ in range(reps))
            kind = "port"
            what = "C restatement (gcc -O2): 1 morph pass + 1024 x skinning pass"
        result["cpu_baseline"] = {"value": reps * ni * model.nv / secs, "unit": "vertices/s", "cores": 1,
                                  "kind": kind,
                                  "sample": f"{reps} x the full config-3 crowd step ({what}), "
                                            f"{secs:.1f} s of CPU work, single thread as the reference runs"}
        # The same baseline on all host cores (one independent Poser per thread, the crowd's
        # instances split among them) -- an extra, the contract's cpu_baseline stays the single-thread figure.
        try:
            if reference_available():
                import concurrent.futures as cf
                import time as _t
                cores = min(len(os.sched_getaffinity(0)), 16)
                per = ni // cores
                with cf.ThreadPoolExecutor(cores) as ex:          # ctypes releases the GIL inside libmmd
                    refs = list(ex.map(lambda _: Reference(model, normalize=True), range(cores)))
                    t0 = _t.perf_counter()
                    list(ex.map(lambda k: [refs[k].time_crowd(rates, pals[k * per:(k + 1) * per]) for _ in range(40)],
                                range(cores)))
                    wall = _t.perf_counter() - t0
                for r_ in refs:
                    r_.close()
                result["cpu_baseline_all_cores"] = {
                    "value": 40 * cores * per * model.nv / wall, "unit": "vertices/s", "cores": cores, "kind": "reference",
                    "sample": f"40 x the config-3 crowd step, {cores} threads x {per} instances each with its own libmmd "
                              f"Poser (every thread repeats the shared morph pass), {wall:.1f} s wall"}
        except Exception as e:                               # pragma: no cover - reporting only
            result["cpu_baseline_all_cores"] = {"error": repr(e)}

    # ---- other BASELINE configs (rank 0, N=1): reported, not the headline --------------------------
    if rank == 0 and world == 1 and not args.no_extras:
        result["other_workloads"] = extras(api, synth, DeformModel, DeviceBuffer, dm, model)

    d_pal.free(); d_w.free(); d_w2.free(); d_a.free()
    if d_b:
        d_b.free()
    dm.close()
    rv.close()
    if rank == 0:
        print(json.dumps(result))


def kernel_source_sha() -> str:
    """Identifies the revision of the code that shapes the deform launch (kernel, launch parameters, plan)."""
    import hashlib
    h = hashlib.sha1()
    for f in ("kernels.hip", "kernels.hpp", "api.cpp", "plan.cpp", "plan.hpp"):
        h.update(open(os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(is_default_layout: bool, ni: int, nv: int, store_policy: str = "nt"):
    """HBM bytes per launch of the deform kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/config3_pmc_hbm_traffic.csv: FETCH_SIZE and WRITE_SIZE collected in separate
    passes).  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts wide streaming reads at
    half their bytes -- the same file's 1 GiB copy_kernel row (FETCH = 524 299 KB) confirms it -- so the
    read side is doubled; WRITE_SIZE is exact.  bench.py cannot drive PMC itself, hence a recorded
    figure with its provenance -- and only while that provenance matches this run: the summary's .meta.json
    (tools/summarize_profiles.py) records the kernel source revision, instance and vertex counts it was
    collected with; on any mismatch the figure is withheld (null) rather than going stale silently."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "config3_pmc_hbm_traffic.csv")))
    if not files or not is_default_layout:
        return {"traffic": None}
    meta_path = files[-1][:-4] + ".meta.json"
    src = os.path.relpath(files[-1], ROOT)
    if not os.path.exists(meta_path):
        return {"traffic": None, "traffic_withheld": f"{src}: no .meta.json (collected before the kernel revision was recorded)"}
    meta = json.load(open(meta_path))
    want = {"kernel_source_sha": kernel_source_sha(), "instances_per_gpu": ni, "vertices": nv}
    diff = {k: (meta.get(k), v) for k, v in want.items() if meta.get(k) != v}
    if diff or any(os.environ.get(k) for k in ("MMDX_GROUP", "MMDX_THREADS", "MMDX_INTERLEAVE", "MMDX_LDS_TARGET")):
        return {"traffic": None, "traffic_withheld": f"{src} was collected for another build / workload: {diff}"}
    fetch = write = None
    # the deform kernel exists in two store flavours (last template argument: write-through); the row of the one THIS run launched
    flavour = ", true>" if store_policy == "sc1 nt" else ", false>"
    for row in csv.DictReader(open(files[-1])):
        if "deform_kernel" in row["kernel"] and flavour in row["kernel"]:
            if row["counter"] == "FETCH_SIZE":
                fetch = float(row["mean_KB"]) * 1024
            elif row["counter"] == "WRITE_SIZE":
                write = float(row["mean_KB"]) * 1024
    if fetch is None or write is None:
        return {"traffic": None, "traffic_withheld": f"{src} holds no launch of the {store_policy}-store flavour this run used"}
    return {"traffic": write + 2 * fetch, "traffic_detail": {
        "write_bytes": write, "fetch_bytes_reported": fetch, "fetch_correction": 2.0, "source": src, "store_flavour": store_policy}}


def fused_pmc_traffic(out):
    """Counter-side HBM traffic of the per-instance-morph workloads ("reported per kernel from rocprofv3 AND from the
    algorithmic figure; both go in the JSON"): WRITE_SIZE + 2 x FETCH_SIZE per launch from the committed separate PMC passes
    (profiles/rNN/fused_gather_pmc.csv, tools/profile_round.sh stage 7; gfx950 FETCH correction as in pmc_traffic), divided by THIS
    run's time per launch.  The launch is identified by its grid size; withheld when the recorded build is not the running one."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "fused_gather_pmc.csv")))
    if not files:
        return
    meta_path = files[-1][:-4] + ".meta.json"
    src = os.path.relpath(files[-1], ROOT)
    if not os.path.exists(meta_path) or json.load(open(meta_path)).get("kernel_source_sha") != kernel_source_sha():
        for k in ("config2_64_frames_per_launch", "config3prime_per_instance_morphs", "config5_64_frames_per_launch_fp16"):
            if k in out and "error" not in out[k]:
                out[k]["pmc_traffic_withheld"] = f"{src} was collected for another build"
        return
    grids = {"config2_64_frames_per_launch": "401408", "config3prime_per_instance_morphs": "3211264",
             "config5_64_frames_per_launch_fp16": "1048576"}
    rows = list(csv.DictReader(open(files[-1])))
    for k, grid in grids.items():
        if k not in out or "error" in out[k]:
            continue
        v = {r["counter"]: float(r["mean"]) * 1024 for r in rows if "deform_kernel" in r["kernel"] and r["grid_threads"] == grid
             and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE")}
        if len(v) != 2:
            continue
        traffic = v["WRITE_SIZE"] + 2 * v["FETCH_SIZE"]
        ms = out[k]["ms_per_call"]
        out[k].update({"pmc_traffic_bytes": traffic, "pmc_GBs": traffic / (ms * 1e-3) / 1e9,
                       "pmc_frac_of_8TBs": traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "pmc_traffic_over_algorithmic": traffic / (out[k]["algorithmic_GBs"] * 1e9 * ms * 1e-3),
                       "pmc_source": src})
        # Which roof binds: these kernels are bit-exact, i.e. unfused multiply-adds, and their instruction stream is as much a
        # floor as their bytes.  Floors from the same counter passes: HBM = algorithmic bytes at the 8 TB/s peak; VALU = vector
        # wave-instructions x 4 cycles (a wave64 instruction occupies its SIMD-32 for 2 passes x 2: packed f32 math, which is most
        # of the stream) on 1024 SIMDs at 2.4 GHz = 614.4 G wave-instructions/s; LDS = cycles the LDS arrays were busy, per CU.
        sq = {r["counter"]: float(r["mean"]) for r in rows if "deform_kernel" in r["kernel"] and r["grid_threads"] == grid
              and r["counter"] in ("SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE")}
        if len(sq) == 2:
            alg_bytes = out[k]["algorithmic_GBs"] * 1e9 * ms * 1e-3
            floors = {"hbm": alg_bytes / (HBM_PEAK_GBS * 1e9) * 1e6,
                      "valu": sq["SQ_INSTS_VALU"] / 614.4e9 * 1e6,
                      "lds": sq["SQ_LDS_IDX_ACTIVE"] / 256 / 2.4e9 * 1e6}
            bound = max(floors, key=floors.get)
            out[k].update({"floors_us": {n: round(t, 2) for n, t in floors.items()}, "bound": bound,
                           "frac_of_binding_floor": floors[bound] / (ms * 1e3),
                           "valu_wave_instructions": sq["SQ_INSTS_VALU"]})


def time_calls(dm, fn, iters, warm=3, settle_ms=60.0):
    """Average ms per call over `iters` back-to-back calls, after `warm` calls and -- like the headline
    measurement -- after the clock transient that follows an idle period: untimed batches until two
    consecutive batch averages agree within 2 % or `settle_ms` of GPU time has been spent."""
    for _ in range(warm):
        fn()
    dm.sync()
    spent, prev, stable = 0.0, None, 0
    while spent < settle_ms and stable < 2:
        dm.timer_start()
        for _ in range(iters):
            fn()
        t = dm.timer_stop()
        spent += t
        stable = stable + 1 if prev is not None and abs(t / prev - 1) < 0.02 else 0
        prev = t
    dm.timer_start()
    for _ in range(iters):
        fn()
    return dm.timer_stop() / iters


def extras(api, synth, DeformModel, DeviceBuffer, dm3, model3):
    out = {}
    flags_dev = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE

    # config 2: single 50k model, 200 active morphs -- per-launch latency and 64 frames per launch
    nfr = 64
    frames = np.arange(nfr)
    pals = synth.make_palettes(model3, frames)
    rates = synth.morph_weights(model3.nm, frames)
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    d_a, d_b, _pl = dm3.alloc_outputs(api.OUT_SOA, nfr, 16)
    ms1 = time_calls(dm3, lambda: dm3.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr,
                                                         api.OUT_SOA, flags_dev), 200)
    ms64 = time_calls(dm3, lambda: dm3.deform_batched_raw(nfr, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr,
                                                          api.OUT_SOA, flags_dev), 50)
    i = dm3.info
    static = model3.nv * 25 + i.n_bdef1 * 2 + i.n_bdef2 * 8 + i.n_bdef4 * 24
    table = i.n_entries * 16
    b1 = static + table + model3.nv * 24 + model3.nb * 48 + model3.nm * 4
    b64 = static + table + nfr * (model3.nv * 24 + model3.nb * 48 + model3.nm * 4)
    out["config2_single_frame"] = {"ms_per_call": ms1, "vertices_per_s": model3.nv / (ms1 * 1e-3),
                                   "algorithmic_GBs": b1 / (ms1 * 1e-3) / 1e9,
                                   "note": "one 50k-vert frame per launch (frame kernel): bound by the latency chain of one workgroup, not by bandwidth"}
    out["config2_64_frames_per_launch"] = {"ms_per_call": ms64,
                                           "vertices_per_s": nfr * model3.nv / (ms64 * 1e-3),
                                           "algorithmic_GBs": b64 / (ms64 * 1e-3) / 1e9,
                                           "frac_of_8TBs": b64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the same single-frame call recorded into a HIP graph, 64 frames per replay (host cost per frame: one launch per
    # 64 frames instead of 64; the device-side time per frame is the kernel's either way)
    try:
        if os.environ.get("MMDX_BENCH_NO_GRAPH"):        # rocprofv3's tracer crashes on stream capture (ROCm 7.2): the profile
            raise RuntimeError("skipped: MMDX_BENCH_NO_GRAPH")   # runs of tools/profile_round.sh leave this leg out
        import time as _t
        dm3.sync()
        dm3.graph_begin()
        for f in range(nfr):
            dm3.deform_batched_raw(1, d_w.ptr + f * model3.nm * 4, d_pal.ptr + f * model3.nb * 64,
                                   d_a.ptr + f * model3.nv * 12, d_b.ptr + f * model3.nv * 12, api.OUT_SOA, flags_dev)
        g = dm3.graph_end()

        def eager64():
            for f in range(nfr):
                dm3.deform_batched_raw(1, d_w.ptr + f * model3.nm * 4, d_pal.ptr + f * model3.nb * 64,
                                       d_a.ptr + f * model3.nv * 12, d_b.ptr + f * model3.nv * 12, api.OUT_SOA, flags_dev)
        ms_g = time_calls(dm3, g.launch, 20) / nfr
        ms_e = time_calls(dm3, eager64, 20) / nfr

        def wall(fn, n=50):
            dm3.sync()
            t0 = _t.perf_counter()
            for _ in range(n):
                fn()
            host = (_t.perf_counter() - t0) / n
            dm3.sync()
            return host
        out["config2_single_frame"].update({"graph_replay_ms_per_frame": ms_g, "eager_ms_per_frame_64_in_a_row": ms_e,
                                            "host_us_per_frame_graph": wall(g.launch) / nfr * 1e6,
                                            "host_us_per_frame_eager": wall(eager64) / nfr * 1e6})
        g.close()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["config2_single_frame"]["graph_error"] = repr(e)
    for b in (d_pal, d_w, d_a, d_b):
        b.free()
    cpu_reference_frame(out["config2_single_frame"], model3, rates, pals)

    # config 3 with the viewer's interleaved 32-byte vertex as output (SURVEY 8d asks for it next to SoA):
    # positions x 0.1, normals, uv passthrough -- Deform + UpdateDeformedVertices in one kernel.
    try:
        ni = 1024
        d_w3 = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, 30)[0])
        d_pal3 = DeviceBuffer.from_numpy(synth.make_palettes(model3, (np.arange(ni) * 3) % 1801))
        d_v32, _none, _pl = dm3.alloc_outputs(api.OUT_VERTEX32, ni, 32)
        dm3.profile_enable(True)
        ms_v = time_calls(dm3, lambda: dm3.deform_batched_raw(ni, d_w3.ptr, d_pal3.ptr, d_v32.ptr, None, api.OUT_VERTEX32,
                                                              flags_dev | api.WEIGHTS_SHARED, 0.1), 20)
        ncalls, skin_ms, _m = dm3.profile_collect()
        dm3.profile_enable(False)
        bv = static + model3.nv * 8 + model3.nv * 12 + ni * (model3.nv * 32 + model3.nb * 48)
        out["config3_vertex32_output"] = {"ms_per_call": ms_v, "vertices_per_s": ni * model3.nv / (ms_v * 1e-3),
                                          "deform_kernel_ms": skin_ms / ncalls,
                                          "algorithmic_GBs": bv / (skin_ms / ncalls * 1e-3) / 1e9,
                                          "frac_of_8TBs": bv / (skin_ms / ncalls * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "placement": _pl}
        for b in (d_w3, d_pal3, d_v32):
            b.free()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["config3_vertex32_output"] = {"error": repr(e)}

    # config 3 "bucketed" (SURVEY 8d): the model with its vertices pre-sorted by deform type inside each tile, so the
    # class-sorted lanes write their own slots of the output image instead of scattering through a permutation.
    try:
        ni = 1024
        sorted3 = synth.presort_by_class(model3)
        dms = DeformModel(sorted3)
        d_w3 = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, 30)[0])
        d_pal3 = DeviceBuffer.from_numpy(synth.make_palettes(model3, (np.arange(ni) * 3) % 1801))
        d_as, d_bs, _pl = dms.alloc_outputs(api.OUT_SOA, ni, 32)
        dms.profile_enable(True)
        ms_s = time_calls(dms, lambda: dms.deform_batched_raw(ni, d_w3.ptr, d_pal3.ptr, d_as.ptr, d_bs.ptr, api.OUT_SOA,
                                                              flags_dev | api.WEIGHTS_SHARED), 20)
        ncalls, skin_ms, _m = dms.profile_collect()
        dms.profile_enable(False)
        out["config3_bucketed_vertices"] = {"ms_per_call": ms_s, "vertices_per_s": ni * model3.nv / (ms_s * 1e-3),
                                            "deform_kernel_ms": skin_ms / ncalls, "placement": _pl,
                                            "note": "vertices pre-sorted by deform type: same kernel time as file "
                                                    "order -- the LDS scatter is off the critical path"}
        for b in (d_w3, d_pal3, d_as, d_bs):
            b.free()
        dms.close()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["config3_bucketed_vertices"] = {"error": repr(e)}

    # config 3' (SURVEY 8d): the same 1024-instance crowd with PER-INSTANCE morph weights (every instance its
    # own facial state): the fused gather path, 4 instances per pass over a vertex's morph row.
    try:
        ni = 1024
        fr_i = (np.arange(ni) * 7) % 600
        d_wi = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, fr_i))
        d_pali = DeviceBuffer.from_numpy(synth.make_palettes(model3, fr_i))
        d_ai, d_bi, _pl = dm3.alloc_outputs(api.OUT_SOA, ni, 16)
        ms_i = time_calls(dm3, lambda: dm3.deform_batched_raw(ni, d_wi.ptr, d_pali.ptr, d_ai.ptr, d_bi.ptr, api.OUT_SOA,
                                                              flags_dev), 10)
        bi = static + table + ni * (model3.nv * 24 + model3.nb * 48 + model3.nm * 4)
        out["config3prime_per_instance_morphs"] = {"ms_per_call": ms_i, "vertices_per_s": ni * model3.nv / (ms_i * 1e-3),
                                                   "algorithmic_GBs": bi / (ms_i * 1e-3) / 1e9,
                                                   "frac_of_8TBs": bi / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS}
        for b in (d_wi, d_pali, d_ai, d_bi):
            b.free()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["config3prime_per_instance_morphs"] = {"error": repr(e)}

    # OPT-IN, NOT the headline and NOT bit-exact: the same workloads on a model created with MMDX_CREATE_FAST_MATH (multiply-adds
    # contracted; results within the tolerance stated in include/mmdx.h, tests/test_fast_math.py).  What the bit-exact contract costs.
    try:
        ni = 1024
        dmf = DeformModel(model3, fast_math=True)
        fr_i = (np.arange(ni) * 7) % 600
        d_wi = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, fr_i))
        d_pali = DeviceBuffer.from_numpy(synth.make_palettes(model3, fr_i))
        d_ws = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, 30)[0])
        d_af, d_bf, _pl = dmf.alloc_outputs(api.OUT_SOA, ni, 32)
        shared = flags_dev | api.WEIGHTS_SHARED
        ms_step = time_calls(dmf, lambda: dmf.deform_batched_raw(ni, d_ws.ptr, d_pali.ptr, d_af.ptr, d_bf.ptr, api.OUT_SOA, shared), 20)
        ms_k = time_calls(dmf, lambda: dmf.deform_batched_raw(ni, d_ws.ptr, d_pali.ptr, d_af.ptr, d_bf.ptr, api.OUT_SOA,
                                                              shared | api.MORPH_UNCHANGED), 20)
        ms_i = time_calls(dmf, lambda: dmf.deform_batched_raw(ni, d_wi.ptr, d_pali.ptr, d_af.ptr, d_bf.ptr, api.OUT_SOA, flags_dev), 10)
        ms_64 = time_calls(dmf, lambda: dmf.deform_batched_raw(64, d_wi.ptr, d_pali.ptr, d_af.ptr, d_bf.ptr, api.OUT_SOA, flags_dev), 50)
        kbytes, _sb = algorithmic_bytes_config3(model3.nv, model3.nb, model3.nm, i.n_entries, ni, i.n_bdef1, i.n_bdef2, i.n_bdef4)
        bi = static + table + ni * (model3.nv * 24 + model3.nb * 48 + model3.nm * 4)
        b64f = static + table + 64 * (model3.nv * 24 + model3.nb * 48 + model3.nm * 4)
        out["fast_math_opt_in"] = {
            "note": "MMDX_CREATE_FAST_MATH: contracted multiply-adds, within the stated tolerance of the reference, NOT bit-exact; "
                    "the default (everything else in this line) is bit-exact",
            "config3_ms_per_step": ms_step, "config3_vertices_per_s": ni * model3.nv / (ms_step * 1e-3),
            "config3_deform_kernel_ms": ms_k, "config3_deform_kernel_frac_of_8TBs": kbytes / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "config3prime_ms_per_call": ms_i, "config3prime_frac_of_8TBs": bi / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "config2_64_frames_ms_per_call": ms_64, "config2_64_frames_frac_of_8TBs": b64f / (ms_64 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "placement": _pl}
        for b in (d_wi, d_pali, d_ws, d_af, d_bf):
            b.free()
        dmf.close()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["fast_math_opt_in"] = {"error": repr(e)}

    # OPT-IN, NOT the headline: MMDX_CREATE_TILE_ORDER -- the same values, written in the engine's vertex order (the renderer remaps
    # its index buffer once): no LDS image, no per-instance barrier.  Bit-exact modulo the permutation (tests/test_tile_order.py).
    # Measured on ONE pair of output arrays against the default model, interleaved; and together with the fast-math opt-in.
    try:
        ni = 1024
        d_pali = DeviceBuffer.from_numpy(synth.make_palettes(model3, (np.arange(ni) * 3) % 1801))
        d_ws = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, 30)[0])
        d_wi = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, (np.arange(ni) * 7) % 600))
        d_at, d_bt, _pl = dm3.alloc_outputs(api.OUT_SOA, ni, 32)
        shared = flags_dev | api.WEIGHTS_SHARED
        res = {}
        for name, kw in (("default", None), ("tile_order", dict(tile_order=True)), ("tile_order_fast_math", dict(tile_order=True, fast_math=True))):
            dmx = dm3 if kw is None else DeformModel(model3, **kw)
            t_step = time_calls(dmx, lambda: dmx.deform_batched_raw(ni, d_ws.ptr, d_pali.ptr, d_at.ptr, d_bt.ptr, api.OUT_SOA, shared), 20)
            t_k = time_calls(dmx, lambda: dmx.deform_batched_raw(ni, d_ws.ptr, d_pali.ptr, d_at.ptr, d_bt.ptr, api.OUT_SOA,
                                                                 shared | api.MORPH_UNCHANGED), 20)
            t_i = time_calls(dmx, lambda: dmx.deform_batched_raw(ni, d_wi.ptr, d_pali.ptr, d_at.ptr, d_bt.ptr, api.OUT_SOA, flags_dev), 10)
            res[name] = (t_step, t_k, t_i)
            if kw is not None:
                dmx.close()
        kbytes, _sb = algorithmic_bytes_config3(model3.nv, model3.nb, model3.nm, i.n_entries, ni, i.n_bdef1, i.n_bdef2, i.n_bdef4)
        bi = static + table + ni * (model3.nv * 24 + model3.nb * 48 + model3.nm * 4)
        out["tile_order_opt_in"] = {
            "note": "MMDX_CREATE_TILE_ORDER: same values in the engine's vertex order (index buffer remapped once by the caller); measured "
                    "interleaved with the default model on the same output arrays; into plainly allocated arrays in the slow store mode "
                    "the shared-morph crowd can be SLOWER than the default (DESIGN.md 4)",
            "placement": _pl}
        for name, (t_step, t_k, t_i) in res.items():
            out["tile_order_opt_in"][name] = {"config3_ms_per_step": t_step, "config3_deform_kernel_ms": t_k,
                                              "config3_deform_kernel_frac_of_8TBs": kbytes / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "config3prime_ms_per_call": t_i,
                                              "config3prime_frac_of_8TBs": bi / (t_i * 1e-3) / 1e9 / HBM_PEAK_GBS}
        for b in (d_pali, d_ws, d_wi, d_at, d_bt):
            b.free()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["tile_order_opt_in"] = {"error": repr(e)}

    # The palette producer for the crowd: a 300-track bone motion with Bezier curves
    # -> local poses -> FK palettes for 1024 instances at their own frames, all in HBM; then the whole
    # motion -> vertices step (poses + palettes + morph pass + deform).  CPU: libmmd doing the same bone
    # work per instance (GetBonePose/SetBonePose per bone + Pre/PostPhysicsPosing), one thread.
    try:
        from simple_mmd_renderer_amd import vmd as vmdmod
        ni = 1024
        names = [f"b{i}" for i in range(model3.nb)]
        vm = vmdmod.Vmd(vmdmod.write_vmd(synth.make_bone_keys(names, 303, keys_per=20, span=600), []))
        bm = vm.bind_bones(names)
        sk = vmdmod.Skeleton(model3.bone_pos, np.asarray(model3.bone_parent, np.int32))
        fr_i = ((np.arange(ni) * 7) % 600).astype(np.uint32)
        d_fr = DeviceBuffer.from_numpy(fr_i)
        d_pose, d_pal = DeviceBuffer(ni * model3.nb * 32), DeviceBuffer(ni * model3.nb * 64)
        d_w = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, 3))
        sa, sb = dm3.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)

        def producer():
            bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm3)
            sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm3)

        def whole():
            producer()
            dm3.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags_dev | api.WEIGHTS_SHARED)
        ms_p, ms_w = time_calls(dm3, producer, 50), time_calls(dm3, whole, 20)
        # the same as ONE call (mmdx_skeleton_solve_motion: a workgroup per instance keeps the poses in LDS, one launch)
        ms_p1 = time_calls(dm3, lambda: sk.solve_motion_device(bm, ni, d_fr.ptr, d_pal.ptr, dm3), 50)
        c = {"instances": ni, "bones": model3.nb, "bone_keys": bm.n_keys, "curve_tables": bm.n_curves,
             "gpu_ms_poses_plus_palettes": ms_p, "gpu_palettes_per_s": ni * model3.nb / (ms_p * 1e-3),
             "gpu_ms_tracks_to_palettes_one_call": ms_p1, "gpu_palettes_per_s_one_call": ni * model3.nb / (ms_p1 * 1e-3),
             "gpu_ms_motion_to_vertices": ms_w, "gpu_vertices_per_s": ni * model3.nv / (ms_w * 1e-3)}
        from oracle.pyoracle import Reference, ReferenceMotion, reference_available   # checker, CPU leg only
        if reference_available():
            import tempfile
            path = os.path.join(tempfile.mkdtemp(prefix="mmdx_"), "crowd.vmd")
            open(path, "wb").write(vmdmod.write_vmd(synth.make_bone_keys(names, 303, keys_per=20, span=600), []))
            rmot = ReferenceMotion(path)
            rsk = Reference.skeleton(model3.bone_pos, np.asarray(model3.bone_parent, np.int64))
            secs = rmot.time_motion_solve(rsk, fr_i[:256]) * (ni / 256)
            c.update({"cpu_reference_ms_poses_plus_palettes": secs * 1e3,
                      "cpu_reference_palettes_per_s": ni * model3.nb / secs})
            rmot.close(); rsk.close()
        out["config3_motion_to_palettes"] = c
        # ONE model, ONE frame, the whole of the viewer's frame() on the device (main.cpp:1786-1825 minus physics and drawing): bone
        # tracks -> local poses -> FK palette (one launch) -> morph gather + skinning (one launch), everything resident in HBM; one
        # submission as a HIP-graph replay (64 frames recorded back to back).  libmmd's whole frame for this model is
        # config2_single_frame.cpu_reference_ms_per_frame.
        try:
            d_fr1 = DeviceBuffer.from_numpy(np.arange(64, dtype=np.uint32) * 3)
            d_w1 = DeviceBuffer.from_numpy(synth.morph_weights(model3.nm, np.arange(64)))

            def frame(f):
                sk.solve_motion_device(bm, 1, d_fr1.ptr + 4 * f, d_pal.ptr, dm3)      # bone tracks -> palette, one launch
                dm3.deform_batched_raw(1, d_w1.ptr + f * model3.nm * 4, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags_dev)

            def eager64():
                for f in range(64):
                    frame(f)
            ms_f = time_calls(dm3, eager64, 10) / 64
            fr = {"gpu_ms_per_frame_eager": ms_f, "launches_per_frame": 2}
            if not os.environ.get("MMDX_BENCH_NO_GRAPH"):
                dm3.sync()
                dm3.graph_begin()
                eager64()
                g = dm3.graph_end()
                fr["gpu_ms_per_frame_graph_replay"] = time_calls(dm3, g.launch, 10) / 64
                g.close()
            out["config2_whole_frame_on_device"] = fr
            d_fr1.free(); d_w1.free()
        except Exception as e:                               # pragma: no cover - reporting only
            out["config2_whole_frame_on_device"] = {"error": repr(e)}
        # the same crowd on a rig with CCD-IK chains and append bones: the ordered solver (the reference's
        # evaluation sequence cut into rounds of independent bones / IK solves).  One of the 8 chains asks for
        # 300 iterations (clamped to the reference's 256) over 3 limited links: that chain alone is the
        # critical path.  Also the rig with the append bones only.
        rig = synth.make_ik_rig(model3.nb, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)
        ski = vmdmod.Skeleton(*rig)
        ska = vmdmod.Skeleton(rig[0], rig[1], rig[2], (np.asarray(rig[3]) & ~np.uint16(0x20)).astype(np.uint16), rig[4], rig[5])
        ms_a = time_calls(dm3, lambda: (bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm3),
                                        ska.solve_device(ni, d_pose.ptr, d_pal.ptr, dm3)), 20)
        out["config3_append_rig_palettes"] = {"instances": ni, "bones": model3.nb, "append_bones": ska.info["n_append_bones"],
                                              "solve_rounds": ska.info["n_solve_rounds"], "gpu_ms_poses_plus_palettes": ms_a,
                                              "gpu_palettes_per_s": ni * model3.nb / (ms_a * 1e-3)}

        def producer_ik():
            bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm3)
            ski.solve_device(ni, d_pose.ptr, d_pal.ptr, dm3)
        ms_i = time_calls(dm3, producer_ik, 10)
        ci = {"instances": ni, "bones": model3.nb, "ik_chains": ski.info["n_ik_bones"], "ik_links": ski.info["n_ik_links"],
              "append_bones": ski.info["n_append_bones"], "solve_rounds": ski.info["n_solve_rounds"],
              "gpu_ms_poses_plus_palettes": ms_i,
              "gpu_palettes_per_s": ni * model3.nb / (ms_i * 1e-3)}
        if reference_available():
            rmot = ReferenceMotion(path)
            rski = Reference.skeleton(rig[0], np.asarray(rig[1], np.int64), rig[2], rig[3], np.asarray(rig[4], np.int64), rig[5],
                                      dict(rig[6], target=np.asarray(rig[6]["target"], np.int64),
                                           link_bone=np.asarray(rig[6]["link_bone"], np.int64)))
            secs = rmot.time_motion_solve(rski, fr_i[:128]) * (ni / 128)
            ci.update({"cpu_reference_ms_poses_plus_palettes": secs * 1e3,
                       "cpu_reference_palettes_per_s": ni * model3.nb / secs})
            rmot.close(); rski.close()
        out["config3_ik_rig_palettes"] = ci
        for b in (d_fr, d_pose, d_pal, d_w, d_a, d_b):
            b.free()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["config3_motion_to_palettes"] = {"error": repr(e)}

    # config 1: 20k verts / 150 bones / 30 morphs, 600 frames -- the reference's own CPU-runnable case.
    # GPU: the 600 frames as 6 batched calls of 100 (per-frame morph weights, fused gather), interleaved
    # 32-byte output incl. the 0.1 scale = Deform + UpdateDeformedVertices.  CPU: libmmd's whole frame
    # (ResetPosing, SetMorphPose x NM, Pre/PostPhysicsPosing, Deform, repack) and the C restatement.
    m1 = synth.make_config("config1_20k")
    dm1 = DeformModel(m1)
    fr = np.arange(600)
    pals1, rates1 = synth.make_palettes(m1, fr), synth.morph_weights(m1.nm, fr)
    d_pal, d_w = DeviceBuffer.from_numpy(pals1), DeviceBuffer.from_numpy(rates1)
    sa, _ = dm1.out_sizes(api.OUT_VERTEX32, 100)
    d_a = DeviceBuffer(sa)

    def frames600():
        for b in range(6):
            dm1.deform_batched_raw(100, d_w.ptr + b * 100 * m1.nm * 4, d_pal.ptr + b * 100 * m1.nb * 64,
                                   d_a.ptr, None, api.OUT_VERTEX32, flags_dev, 0.1)
    ms600 = time_calls(dm1, frames600, 10)
    c1 = {"gpu_ms_per_600_frames": ms600, "gpu_vertices_per_s": 600 * m1.nv / (ms600 * 1e-3)}
    try:
        from oracle.pyoracle import Oracle, Reference, reference_available   # checker, CPU leg only
        import time as _t
        orc = Oracle()
        skin = orc.normalize(m1)
        t0 = _t.perf_counter()
        for f in range(600):
            pos, nrm = orc.skin(m1, pals1[f], orc.morph(m1, rates1[f]), skin)
            orc.repack32(m1, pos, nrm, 0.1)
        c1["cpu_port_vertices_per_s"] = 600 * m1.nv / (_t.perf_counter() - t0)
        if reference_available():
            ref = Reference(m1, normalize=True)
            c1["cpu_reference_vertices_per_s"] = 600 * m1.nv / ref.time_frames(rates1, pals1)
            ref.close()
    except Exception as e:                                   # pragma: no cover - reporting only
        c1["cpu_error"] = repr(e)
    out["config1_600_frames"] = c1
    for b in (d_pal, d_w, d_a):
        b.free()

    # The drop-in call as the viewer would make it: host palette + rates in, 32-byte vertices out to host
    # memory (PCIe inclusive, synchronous) -- pageable numpy buffers vs page-locked ones.
    try:
        import ctypes as C
        from simple_mmd_renderer_amd.engine import PinnedArray
        pal1, rt1 = pals1[7].copy(), rates1[7].copy()
        out_pg = np.empty((m1.nv, 8), np.float32)
        f32p = C.POINTER(C.c_float)

        def frame(pal, rt, out):
            api.check(api.lib().mmdx_deform_vertex32(dm1.h, rt.ctypes.data_as(f32p), pal.ctypes.data_as(f32p),
                                                     C.c_float(0.1), out.ctypes.data))
        import time as _t
        def wall(fn, n=200):
            for _ in range(10):
                fn()
            t0 = _t.perf_counter()
            for _ in range(n):
                fn()
            return (_t.perf_counter() - t0) / n * 1e3
        pg = wall(lambda: frame(pal1, rt1, out_pg))
        p_pal, p_rt, p_out = PinnedArray(pal1.shape, np.float32), PinnedArray(rt1.shape, np.float32), PinnedArray((m1.nv, 8), np.float32)
        p_pal.array[:] = pal1; p_rt.array[:] = rt1
        pn = wall(lambda: frame(p_pal.array, p_rt.array, p_out.array))
        assert np.array_equal(p_out.array.view(np.uint32), out_pg.view(np.uint32))
        out["config1_frame_host_io"] = {"vertices": m1.nv, "pageable_ms": pg, "pinned_ms": pn,
                                        "pinned_vertices_per_s": m1.nv / (pn * 1e-3),
                                        "note": "mmdx_deform_vertex32, host in / host out, PCIe + sync included"}
        for x in (p_pal, p_rt, p_out):
            x.free()
    except Exception as e:                                   # pragma: no cover - reporting only
        out["config1_frame_host_io"] = {"error": repr(e)}

    # PMX loader (the first "next" row): the config-1 model written as a PMX 2.0 file, parsed by this
    # repo's loader (C++, csrc/pmx.cpp) and by the reference's FileReader + PmxReader -- host CPU both.
    try:
        import tempfile
        import time as _t
        from simple_mmd_renderer_amd import pmx as pmxmod
        data = pmxmod.write_pmx(m1)
        path = os.path.join(tempfile.mkdtemp(prefix="mmdx_"), "config1.pmx")
        open(path, "wb").write(data)
        import ctypes as C
        reps = 20
        t0 = _t.perf_counter()
        for _ in range(reps):
            h = C.c_void_p()
            api.check(api.lib().mmdx_pmx_load_file(path.encode(), C.byref(h)))
            api.lib().mmdx_pmx_destroy(h)
        ours = (_t.perf_counter() - t0) / reps
        c = {"file_MB": len(data) / 1e6, "mmdx_ms": ours * 1e3, "mmdx_MBps": len(data) / 1e6 / ours}
        from oracle.pyoracle import Reference, reference_available
        if reference_available():
            ref_s = Reference.time_pmx_load(path, reps)
            c.update({"reference_ms": ref_s * 1e3, "reference_MBps": len(data) / 1e6 / ref_s})
        out["pmx_loader_config1_file"] = c
    except Exception as e:                                   # pragma: no cover - reporting only
        out["pmx_loader_config1_file"] = {"error": repr(e)}

    dm1.close()

    # config 5: 262 144 verts, 512 bones, 1024 morphs x 4096 entries, fp16 positions
    m5 = synth.make_config("config5_256k")
    dm5 = DeformModel(m5, f16_positions=True)
    pals = synth.make_palettes(m5, frames)
    rates = synth.morph_weights(m5.nm, frames)
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    d_a, d_b, _pl = dm5.alloc_outputs(api.OUT_SOA_POS16, nfr, 16)
    ms1 = time_calls(dm5, lambda: dm5.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr,
                                                         api.OUT_SOA_POS16, flags_dev), 100)
    ms64 = time_calls(dm5, lambda: dm5.deform_batched_raw(nfr, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr,
                                                          api.OUT_SOA_POS16, flags_dev), 20)
    i = dm5.info
    static = m5.nv * (6 + 12 + 1) + i.n_bdef1 * 2 + i.n_bdef2 * 8 + i.n_bdef4 * 24
    table = i.n_entries * 10
    b1 = static + table + m5.nv * 18 + m5.nb * 48 + m5.nm * 4
    b64 = static + table + nfr * (m5.nv * 18 + m5.nb * 48 + m5.nm * 4)
    out["config5_single_frame_fp16"] = {"ms_per_call": ms1, "vertices_per_s": m5.nv / (ms1 * 1e-3),
                                        "algorithmic_GBs": b1 / (ms1 * 1e-3) / 1e9}
    out["config5_64_frames_per_launch_fp16"] = {"ms_per_call": ms64,
                                                "vertices_per_s": nfr * m5.nv / (ms64 * 1e-3),
                                                "algorithmic_GBs": b64 / (ms64 * 1e-3) / 1e9,
                                                "frac_of_8TBs": b64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS}
    for b in (d_pal, d_w, d_a, d_b):
        b.free()
    dm5.close()
    cpu_reference_frame(out["config5_single_frame_fp16"], m5, rates, pals)
    fused_pmc_traffic(out)
    return out


def cpu_reference_frame(entry, model, rates, pals, nframes=4):
    """libmmd's whole frame (ResetPosing, SetMorphPose x NM, Pre/PostPhysicsPosing, Deform, 32-byte repack), one
    thread, f32, next to a single-model GPU figure."""
    try:
        from oracle.pyoracle import Reference, reference_available   # checker, CPU leg only
        if reference_available():
            ref = Reference(model, normalize=True)
            secs = ref.time_frames(rates[:nframes], pals[:nframes]) / nframes
            ref.close()
            entry["cpu_reference_ms_per_frame"] = secs * 1e3
            entry["cpu_reference_vertices_per_s"] = model.nv / secs
    except Exception as e:                                   # pragma: no cover - reporting only
        entry["cpu_error"] = repr(e)


if __name__ == "__main__":
    main()
