#!/bin/bash
# Round 3, GPU call 28: CCD-IK solver, plain (292 VGPRs, 1 workgroup per CU) vs dense (256 VGPRs, 2 per CU) at every crowd size.
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_rig.py tests/test_physics_seam.py tests/test_bullet_reactor.py -m gpu -x -q 2>&1 | tail -2
for d in 0 1 auto; do
  if [ $d = auto ]; then unset MMDX_SOLVE_DENSE; else export MMDX_SOLVE_DENSE=$d; fi
  echo "== MMDX_SOLVE_DENSE=$d" | tee -a $out/ik_dense_ab.txt
  RIG_ONLY=ik RIG_NI=1024,2048,4096,8192,16384 timeout -k 10 300 python tools/rig_bench.py 2>&1 | tee -a $out/ik_dense_ab.txt
done
MMDX_SOLVE_DENSE=1 timeout -k 10 300 python tools/soak_rig.py 300 128 2>&1 | tail -1
