#!/bin/bash
# Round 3, GPU call 54: ordered solver with the next round's event record requested ahead: parity + timing.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_rig.py tests/test_physics_seam.py tests/test_bullet_reactor.py tests/test_graph.py -m gpu -x -q 2>&1 | tail -2
RIG_NI=1024,16384 timeout -k 10 300 python tools/rig_bench.py 2>&1 | tee $out/rig_event_prefetch.txt
timeout -k 10 300 python tools/soak_rig.py 500 128 2>&1 | tail -1
