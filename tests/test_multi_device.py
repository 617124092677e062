"""GPU: the PRODUCT on more than one shard -- (a) in one process, one host thread + one
handle / stream per shard; (b) one process per rank through torch.distributed.run with the HIP path as the
deformer; (c) `python bench.py --gpus 2` starting its own ranks.  On the 1-GPU box the shards share device 0
(device = shard % n_devices); on a multi-GPU node the same tests spread over the devices."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth
from simple_mmd_renderer_amd.crowd import InProcessCrowd, crowd_frames
from simple_mmd_renderer_amd.engine import DeformModel, device_count
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    assert device_count() >= 1, "no HIP device visible: the GPU tests must run on the MI355X box"


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("shards", [2, 3])
def test_in_process_shards_one_thread_and_stream_each(oracle, shards):
    m = synth.make_model(6007, 60, 8, 200, seed=808)
    ni = 37
    pals = synth.make_palettes(m, crowd_frames(0, ni))
    rates = synth.morph_weights(m.nm, 30)[0]
    per = synth.morph_weights(m.nm, np.arange(ni) * 5)
    with DeformModel(m) as one:
        want = one.deform_batched(rates, pals, shared_weights=True)
        want_per = one.deform_batched(per, pals)
        want32 = one.deform_batched(rates, pals, shared_weights=True, layout=api.OUT_VERTEX32, pos_scale=0.1)
    crowd = InProcessCrowd(m, shards)
    try:
        assert [mm.info.device_ordinal for mm in crowd.models] == [s % device_count() for s in range(shards)]
        for rep in range(3):                                    # repeated: races show as flaky mismatches
            got = crowd.deform(rates, pals)
            got_per = crowd.deform(per, pals, shared_weights=False)
            got32 = crowd.deform(rates, pals, layout=api.OUT_VERTEX32, pos_scale=0.1)
            for k in range(2):
                assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), f"rep {rep} shared [{k}]"
                assert np.array_equal(got_per[k].view(np.uint32), want_per[k].view(np.uint32)), f"rep {rep} per [{k}]"
            assert np.array_equal(got32.view(np.uint32), want32.view(np.uint32))
    finally:
        crowd.close()
    skin = oracle.normalize(m)
    vimg = oracle.morph(m, rates)
    for i in (0, ni // 2, ni - 1):
        ep, en = oracle.skin(m, pals[i], vimg, skin)
        gu.assert_bits_equal(want[0][i], ep, f"inst {i} pos")
        gu.assert_bits_equal(want[1][i], en, f"inst {i} nrm")


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_run_the_hip_path_on_their_own_devices(oracle, tmp_path, world):
    total = 11
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(out), str(total), "--hip"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.load(open(out))
    assert res["world"] == world and res["n_total"] == total
    assert res["devices"] == [k % device_count() for k in range(world)]
    model = synth.make_model(1500, 40, 6, 100, seed=99)
    rates = synth.morph_weights(model.nm, 30)[0]
    pals = synth.make_palettes(model, crowd_frames(0, total))
    skin = oracle.normalize(model)
    vimg = oracle.morph(model, rates)
    want = []
    for i in range(total):
        pos, nrm = oracle.skin(model, pals[i], vimg, skin)
        want.append(synth.checksum64(np.concatenate([pos.ravel(), nrm.ravel()])))
    assert res["checksums"] == want


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the driver's command shape): one JSON line, n_gpus 2."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--instances-per-gpu", "64", "--no-cpu-baseline", "--no-extras", "--plain-alloc"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
    assert line["config"]["instances_per_gpu"] == 64
    assert line["value"] > 0 and abs(line["value"] - 2 * 64 * 50000 * 5 / (line["ms_per_step"] * 5e-3)) < 1e-6 * line["value"]
    assert 0 < line["roofline"]["frac"] < 1 and 0 < line["roofline"]["step_frac"] < 1
