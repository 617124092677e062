"""Build the in-tree HIP library (libmmdx.so) for gfx950 with hipcc.

    python -m simple_mmd_renderer_amd.build [--force]

One explicit hipcc command, no build system: the product is a dozen translation units (kernels_fast.hip is kernels.hip
compiled a second time with multiply-add contraction allowed, for models created with MMDX_CREATE_FAST_MATH).  The .so is
git-ignored but travels to the GPU box with the working tree.  -ffp-contract=off is part of the
contract (bit-exact parity with the reference's CPU arithmetic), not a debug flag.
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


def flags() -> list[str]:
    extra = []
    if os.environ.get("MMDX_BUILD_DEFS"):      # tools/ experiments only
        extra += ["-D" + d for d in os.environ["MMDX_BUILD_DEFS"].split(",")]
    if os.environ.get("MMDX_BUILD_TILE"):
        extra.append("-DMMDX_TILE=" + os.environ["MMDX_BUILD_TILE"])
    return extra + [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
            "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter"]


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


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
        cmd = [hipcc()] + flags() + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
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
