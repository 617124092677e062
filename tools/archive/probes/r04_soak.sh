f=gpurun_out/r04_soak_final_tree.txt
echo "== 12000 seeds of randomized models x every call form (tests/test_gpu_parity.py::test_randomized_models_all_call_forms)" > $f
MMDX_SOAK_SEEDS=12000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -1 >> $f
echo "== MMDX_FUSED_PACK=1 (per-instance morph weights through pack_kernel), 6000 seeds" >> $f
MMDX_FUSED_PACK=1 MMDX_SOAK_SEEDS=6000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -1 >> $f
echo "== MMDX_STORE_WT=1, 6000 seeds" >> $f
MMDX_STORE_WT=1 MMDX_SOAK_SEEDS=6000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -1 >> $f
echo "== MMDX_MORPH_AUTOSKIP=0, 3000 seeds" >> $f
MMDX_MORPH_AUTOSKIP=0 MMDX_SOAK_SEEDS=3000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized_models_all_call_forms 2>&1 | tail -1 >> $f
echo "== 6000 rigs x 128 instances (device bone solve incl. CCD-IK, sixteen lanes per solve, vs the C oracle; tools/soak_rig.py)" >> $f
timeout -k 10 900 python3 tools/soak_rig.py 6000 128 2>&1 | tail -1 >> $f
echo "== MMDX_IK_COOP=0, 1500 rigs x 128 instances" >> $f
MMDX_IK_COOP=0 timeout -k 10 600 python3 tools/soak_rig.py 1500 128 2>&1 | tail -1 >> $f
cat $f
