#!/bin/bash
# Round 3, GPU call 27: CCD-IK solver with two workgroups per CU (256 VGPRs, 72 KB of LDS windows): parity + scaling.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_rig.py tests/test_physics_seam.py tests/test_bullet_reactor.py -m gpu -x -q 2>&1 | tail -2
RIG_ONLY=ik RIG_NI=1024,4096,8192,16384 timeout -k 10 300 python tools/rig_bench.py 2>&1 | tee $out/ik_two_wg_per_cu.txt
timeout -k 10 300 python tools/soak_rig.py 300 128 2>&1 | tail -1
