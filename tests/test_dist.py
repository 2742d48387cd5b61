"""The sample-split multi-GPU path on CPU: two gloo ranks, the oracle standing in for the device
renderer, one sum-reduce of the accumulators (what RCCL does over xGMI on the GPU box)."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import dist as rdist


def test_frame_stripes_partition_the_frames():
    for total in (0, 1, 5, 8, 1024):
        for world in (1, 2, 3, 8):
            frames = []
            for r in range(world):
                first, count, stride = rdist.frame_stripe(total, r, world)
                frames += [first + k * stride for k in range(count)]
            assert sorted(frames) == list(range(1, total + 1))
    with pytest.raises(ValueError):
        rdist.frame_stripe(4, 2, 2)


def _worker(rank, world, port, total_frames, out_path):
    import torch
    import torch.distributed as dist
    from oracle import orc
    from raytracer_amd import procedural
    from util import oracle_scene
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = procedural.material_test_scene(tex=16, env_size=(32, 16))
    a = s.arrays()
    res = (32, 24)
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    accum = torch.zeros(res[1], res[0], 4, dtype=torch.float32)

    def render_fn(first, count, stride):
        orc.render(oracle_scene(a), u, first_frame=first, n_frames=count, frame_stride=stride, accum=accum.numpy())

    rdist.render_striped(render_fn, accum, total_frames, rank, world,
                         reduce_fn=lambda t: dist.reduce(t, dst=0, op=dist.ReduceOp.SUM))
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_frames", [6, 1])
def test_two_rank_spp_split_equals_single_rank(tmp_path, total_frames):
    import torch.multiprocessing as mp
    from oracle import orc
    from raytracer_amd import procedural
    from util import oracle_scene
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "accum.npy")
    mp.spawn(_worker, args=(2, port, total_frames, out), nprocs=2, join=True)
    got = np.load(out)
    s = procedural.material_test_scene(tex=16, env_size=(32, 16))
    a = s.arrays()
    ref = orc.render(oracle_scene(a), s.frame_uniforms((32, 24)), first_frame=1, n_frames=total_frames)["accum"]
    assert np.array_equal(got[..., 3], ref[..., 3])
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7)
