#!/usr/bin/env python3
"""VGPRs / SGPRs / spills / static LDS of every kernel in a built libmmdx.so (no GPU needed):
    python tools/kernel_resources.py [path.so] [substring ...]
Unbundles the gfx950 code object and reads the AMDGPU metadata note."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "simple_mmd_renderer_amd", "libmmdx.so")
    want = sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        co = os.path.join(d, "gfx950.co")
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\((?!anonymous).*$", "", dem).replace("void ", "").replace("(anonymous namespace)::", "").replace("mmdx::", "")
        if want and not any(w in dem for w in want):
            continue
        print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} "
              f"scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
