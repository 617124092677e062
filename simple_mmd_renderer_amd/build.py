"""Build the in-tree HIP library (libmmdx.so) for gfx950 with hipcc.

    python -m simple_mmd_renderer_amd.build [--force]

Explicit hipcc commands, no build system: one `hipcc -c` per translation unit (a dozen; in parallel; objects cached by content
under build/obj), one link (kernels_fast.hip is kernels.hip compiled a second time with multiply-add contraction allowed, for
models created with MMDX_CREATE_FAST_MATH).  The .so is git-ignored but travels to the GPU box with the working tree and carries
the hash of the sources it was built from (mmdx_build_source_sha): freshness is decided by content, never by timestamps.
-ffp-contract=off is part of the contract (bit-exact parity with the reference's CPU arithmetic), not a debug flag.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("MMDX_BUILD_OUT") or os.path.join(HERE, "libmmdx.so")   # MMDX_BUILD_OUT: experiment builds (tools/)
SOURCES = ["api.cpp", "bench_api.cpp", "kernels.hip", "kernels_fast.hip", "plan.cpp", "pmx.cpp", "pmd.cpp", "vmd.cpp", "error.cpp",
           "rig.cpp", "rig_api.cpp", "rig_kernels.hip"]
HEADERS = ["kernels.hpp", "plan.hpp", "error.hpp", "vmd.hpp", "rig.hpp", "rig_kernels.hpp", "pmx.hpp", "graph_pin.hpp", "api_internal.hpp",
           os.path.join("..", "..", "include", "mmdx.h"), os.path.join("..", "..", "include", "mmdx_bench.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def source_sha() -> str:
    """SHA-1 over everything the library is built from (sources and headers in the order listed, the flags, experiment knobs):
    embedded in the binary (mmdx_build_source_sha) and compared by build(), bench.py and smoke() -- the loaded library is
    provably built from the files in the tree, whatever the timestamps say."""
    import hashlib
    h = hashlib.sha1()
    for f in SOURCES + HEADERS:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(flags()).encode())
    return h.hexdigest()


def library_sha(path: str = None) -> str | None:
    """The stamp inside a built library, read from the file's bytes (no dlopen); None if missing / unstamped."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    tag = b"mmdx-source-sha:"
    i = data.find(tag)
    if i < 0:
        return None
    sha = data[i + len(tag): i + len(tag) + 40]
    return sha.decode() if len(sha) == 40 and all(c in b"0123456789abcdef" for c in sha) else None


STAMPED = "bench_api.cpp"      # the one translation unit that carries -DMMDX_SOURCE_SHA (mmdx_build_source_sha lives there)
INCLUDES = {"kernels_fast.hip": ["kernels.hip"]}      # sources that #include other sources


def flags(stamp: bool = False) -> list[str]:
    extra = []
    if stamp:
        extra.append('-DMMDX_SOURCE_SHA="%s"' % source_sha())
    if os.environ.get("MMDX_BUILD_DEFS"):      # tools/ experiments only
        extra += ["-D" + d for d in os.environ["MMDX_BUILD_DEFS"].split(",")]
    if os.environ.get("MMDX_BUILD_TILE"):
        extra.append("-DMMDX_TILE=" + os.environ["MMDX_BUILD_TILE"])
    return extra + [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
            "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter"]


OBJ_DIR = os.path.join(HERE, "..", "build", "obj")


def _compile_one(args):
    src, obj, cmd = args
    if os.path.exists(obj):
        return obj, 0, ""
    tmp = obj + ".tmp.%d" % os.getpid()
    r = subprocess.run(cmd + ["-c", "-x", "hip", src, "-o", tmp], capture_output=True, text=True)
    if r.returncode == 0:
        os.replace(tmp, obj)
    elif os.path.exists(tmp):
        os.remove(tmp)
    return obj, r.returncode, r.stdout + r.stderr


def _objects(verbose: bool) -> list[str]:
    """One object per translation unit, keyed by the hash of what goes into it -- its source (and sources it #includes), every
    header, the flags -- so a header edit rebuilds all of them, an edit of one source only that one (plus the stamped unit), and
    an experiment's -D never picks up the default build's objects.  Compiled in parallel."""
    import concurrent.futures as cf
    import hashlib
    os.makedirs(OBJ_DIR, exist_ok=True)
    common = hashlib.sha1(" ".join(flags()).encode())
    for f in HEADERS:
        common.update(open(os.path.join(CSRC, f), "rb").read())
    jobs = []
    for src in SOURCES:
        stamped = src == STAMPED
        h = common.copy()
        for f in [src] + INCLUDES.get(src, []):
            h.update(open(os.path.join(CSRC, f), "rb").read())
        if stamped:
            h.update(source_sha().encode())
        jobs.append((os.path.join(CSRC, src), os.path.join(OBJ_DIR, "%s-%s.o" % (os.path.splitext(src)[0], h.hexdigest()[:16])),
                     [hipcc()] + flags(stamp=stamped)))
    cmd = [hipcc()] + flags()
    if verbose:
        print(" ".join(cmd + ["-c", "-x", "hip", "<each of: %s>" % " ".join(SOURCES)]))
    workers = max(1, min(len(jobs), len(os.sched_getaffinity(0))))
    with cf.ThreadPoolExecutor(workers) as ex:
        done = list(ex.map(_compile_one, jobs))
    for obj, rc, log in done:
        if rc != 0:
            sys.stderr.write(log)
            raise RuntimeError("hipcc failed for " + os.path.basename(obj))
        if verbose and log:
            sys.stderr.write(log)
    # older revisions' objects: keep the cache from growing without bound
    keep = {o for o, _, _ in done}
    olds = sorted((f for f in os.listdir(OBJ_DIR) if f.endswith(".o") and os.path.join(OBJ_DIR, f) not in keep),
                  key=lambda f: os.path.getmtime(os.path.join(OBJ_DIR, f)))
    for f in olds[:-4 * len(SOURCES)] if len(olds) > 4 * len(SOURCES) else []:
        os.remove(os.path.join(OBJ_DIR, f))
    return [o for o, _, _ in done]


def up_to_date() -> bool:
    """By content, not by timestamp: the library's embedded stamp equals the hash of the sources in the tree."""
    return library_sha() == source_sha()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return LIB
    # Several ranks of one job may get here at once (bench.py --gpus N on a fresh copy of the tree whose
    # timestamps make the library look stale): one of them builds, into a temporary file that is renamed over
    # the library in one step, the others wait on the lock and find it up to date.
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date():
            return LIB
        tmp = LIB + ".tmp.%d" % os.getpid()
        objs = _objects(verbose)
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed")
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        os.replace(tmp, LIB)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_EXAMPLE = os.path.join(HOST_DIR, "frame_loop_example")


def build_host_example(force: bool = False, name: str = "frame_loop_example") -> str:
    """The C++ host mirror's runnable examples (g++ only: the host side needs no HIP headers):
    frame_loop_example (palette from the host) and motion_example (.pmx/.pmd + .vmd, everything on the GPU)."""
    src = os.path.join(HOST_DIR, name + ".cpp")
    hdr = os.path.join(HOST_DIR, "mmdx_poser.hpp")
    exe = os.path.join(HOST_DIR, name)
    if (not force and os.path.exists(exe) and
            all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in (src, hdr, LIB))):
        return exe
    build()
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", src,
           "-I" + os.path.join(HERE, "..", "include"), "-L" + HERE, "-lmmdx", "-Wl,-rpath,$ORIGIN/..",
           "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed for host/%s.cpp" % name)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host_example(force="--force" in sys.argv))
    print(build_host_example(force="--force" in sys.argv, name="motion_example"))
