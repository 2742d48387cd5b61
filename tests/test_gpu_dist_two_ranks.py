"""The sample-split multi-GPU path with MORE THAN ONE RANK on hardware: two fresh processes share
GPU 0, each renders its stripe of frames (rank r: frames r+1, r+3, ...) through
raytracer_amd.dist.DeviceRenderer into a torch tensor bound with rthip_bind_accum, the accumulators
are summed across the processes, and the root's result is compared with one context rendering all
frames.  (RCCL does not take two ranks on one device, so the sum goes through gloo's all-reduce of
the device tensors; the render, bind, stripe and reject_hist logic is what runs on a multi-GPU node.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import device, procedural

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
RES = (160, 96)


def run_ranks(tmp_path, total_frames, root, world=2):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / f"accum_{total_frames}_{root}.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), str(r), str(world), str(port), str(total_frames), str(root), "0", out],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return np.load(out), np.load(out + ".rgba8.npy")


@pytest.mark.parametrize("total_frames,root", [(7, 0), (6, 1), (1, 0), (1, 1)])
def test_two_processes_on_one_gpu_equal_one_context(tmp_path, total_frames, root):
    """total_frames = 1 leaves rank 1 without a frame: it contributes zeros, also as the root."""
    got, img = run_ranks(tmp_path, total_frames, root)
    s = procedural.material_test_scene(tex=32, env_size=(64, 32))
    c = device.Context(0)
    try:
        c.upload_scene(s.arrays())
        c.set_frame_uniforms(s.frame_uniforms(RES, frame=1, reject_hist=1))
        c.render(1, total_frames, 1, device.default_params(max_bounces=4))
        ref = c.read_accum()
    finally:
        c.close()
    assert np.array_equal(got[..., 3], ref[..., 3]) and np.all(ref[..., 3] == float(total_frames))   # sample counts: exact
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)                                     # sums: another order of additions
    want = orc.resolve(got)
    d = np.abs(img.astype(int) - want.astype(int))
    assert d.max() <= 1
