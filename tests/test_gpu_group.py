"""Several GPUs from one process (include/rthip_group.h), on the GPUs this box has.

With one GPU the group degenerates to a single context plus a one-rank RCCL reduce, which must
change nothing; the striping arithmetic for more ranks is covered on the CPU
(test_cabi.py::test_group_plan_stripes_every_frame_exactly_once) and, with two contexts standing
in for two GPUs, below.
"""
import numpy as np
import pytest

from raytracer_amd import device, group, procedural

pytestmark = pytest.mark.gpu

RES = (256, 144)


@pytest.fixture(scope="module")
def box():
    s = procedural.box_scene()
    return s, s.arrays()


def single(scene, arrays, first, n, stride=1):
    c = device.Context(0)
    try:
        c.upload_scene(arrays)
        c.set_frame_uniforms(scene.frame_uniforms(RES, frame=first, reject_hist=1))
        c.render(first, n, stride, device.default_params())
        return c.read_accum()
    finally:
        c.close()


def test_group_of_all_gpus_equals_one_context(box):
    s, a = box
    n_dev = device.device_count()
    g = group.Group(list(range(n_dev)))
    try:
        assert g.size == n_dev
        g.upload_scene(a)
        g.set_frame_uniforms(s.frame_uniforms(RES, frame=1, reject_hist=1))
        g.render(1, 12, device.default_params())
        g.reduce(0)
        got = g.contexts[0].read_accum()
        ref = single(s, a, 1, 12)
        assert np.all(got[..., 3] == 12.0)
        if n_dev == 1:
            assert np.array_equal(got, ref)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
        # rendering goes on after a reduce; the second reduce adds only the new samples
        g.render(13, 5, device.default_params())
        g.reduce(0)
        got = g.contexts[0].read_accum()
        assert np.all(got[..., 3] == 17.0)
        np.testing.assert_allclose(got, single(s, a, 1, 17), rtol=1e-5, atol=1e-5)
        img = g.contexts[0].resolve()
        assert img.shape == (RES[1], RES[0], 4)
    finally:
        g.close()


def test_group_resize_and_restart(box):
    s, a = box
    g = group.Group([0])
    try:
        g.upload_scene(a)
        g.set_frame_uniforms(s.frame_uniforms(RES, frame=1, reject_hist=1))
        g.render(1, 3, device.default_params())
        small = (64, 48)
        g.set_frame_uniforms(s.frame_uniforms(small, frame=1, reject_hist=1))
        g.reduce(0)  # nothing rendered at the new size yet: an empty accumulator
        assert not g.contexts[0].read_accum().any()
        g.render(1, 2, device.default_params())
        g.reduce(0)
        got = g.contexts[0].read_accum()
        assert got.shape == (48, 64, 4) and np.all(got[..., 3] == 2.0)
    finally:
        g.close()


def test_group_errors_are_reported(box):
    with pytest.raises(device.RtHipError, match="listed twice"):
        group.Group([0, 0])
    with pytest.raises(device.RtHipError):
        group.Group([device.device_count()])
    g = group.Group([0])
    try:
        with pytest.raises(device.RtHipError, match="no frame uniforms"):
            g.render(1, 1)
        s, a = box
        g.upload_scene(a)
        g.set_frame_uniforms(s.frame_uniforms(RES, frame=1, reject_hist=1))
        with pytest.raises(device.RtHipError, match="root"):
            g.reduce(1)
    finally:
        g.close()


def test_two_striped_contexts_hold_the_samples_of_one(box):
    """What two ranks of a group do, with two contexts on this GPU and the sum taken on the host."""
    s, a = box
    parts = []
    for r in range(2):
        f, c = group.plan(1, 9, 2, r)
        parts.append(single(s, a, f, c, stride=2))
    total = parts[0] + parts[1]
    ref = single(s, a, 1, 9)
    assert np.all(total[..., 3] == 9.0)
    np.testing.assert_allclose(total, ref, rtol=1e-5, atol=1e-5)


def test_group_of_two_gpus_equals_one_context(box):
    """Two real GPUs: frames striped 1,3,5,... and 2,4,6,..., one ncclReduce over xGMI to GPU 1 and
    then to GPU 0; a short render (fewer frames than GPUs) leaves the frame-less GPU able to be root."""
    if device.device_count() < 2:
        pytest.skip("needs two GPUs")
    s, a = box
    g = group.Group([0, 1])
    try:
        g.upload_scene(a)
        g.set_frame_uniforms(s.frame_uniforms(RES, frame=1, reject_hist=1))
        g.render(1, 13, device.default_params())
        g.reduce(0)
        got = g.contexts[0].read_accum()
        np.testing.assert_allclose(got, single(s, a, 1, 13), rtol=1e-5, atol=1e-5)
        assert np.all(got[..., 3] == 13.0)
        # one frame for two GPUs, root = the GPU that got none; then more samples on top
        g.set_frame_uniforms(s.frame_uniforms(RES, frame=1, reject_hist=1))
        g.render(1, 1, device.default_params())
        g.reduce(1)
        g.set_frame_uniforms(s.frame_uniforms(RES, frame=2, reject_hist=0))
        g.render(2, 4, device.default_params())
        g.reduce(1)
        got = g.contexts[1].read_accum()
        np.testing.assert_allclose(got, single(s, a, 1, 5), rtol=1e-5, atol=1e-5)
    finally:
        g.close()
