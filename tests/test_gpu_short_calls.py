"""Frame groups of at most 8 * 2^20 camera paths (four 1080p frames; any group of a small image) run
as one persistent launch in which a lane owns a whole path (k_path); larger ones, and calls with
RTHIP_FLAG_STAGED, go through the staged pipeline (generate / cull / extend / shade per bounce).  Both run the same code per path, so everything they produce must
be equal bit for bit -- accumulators, primary-hit taps, final RNG states, traversal counters -- and
the short path must stand against the oracle on its own.
"""
import numpy as np
import pytest

from oracle import orc
from raytracer_amd import device, procedural
from util import oracle_params, oracle_scene, psnr, tonemapped

pytestmark = pytest.mark.gpu

SCENES = {
    "box": (procedural.box_scene, dict(env_size=(64, 32)), (67, 41)),
    "materials": (procedural.material_test_scene, dict(tex=64, env_size=(128, 64)), (160, 96)),   # MASK + BLEND, two UV sets, instancing
    "terrain": (procedural.terrain_scene, dict(env_size=(128, 64)), (128, 72)),
    "atrium": (procedural.atrium_scene, dict(detail=0.25, tex=64), (320, 180)),                   # 103 primitives: candidate masks
}
EXACT = ("rays", "prims_visited", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests", "texel_fetches", "shade_calls",
         "misses", "kills", "stack_overflows", "paths")


@pytest.fixture(scope="module")
def ctx():
    c = device.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def scenes():
    out = {}
    for name, (fn, kw, res) in SCENES.items():
        s = fn(**kw)
        out[name] = (s, s.arrays(), res)
    return out


def render(ctx, scene, res, calls, flags=0, **kw):
    """calls: list of (first_frame, n_frames).  Returns accumulator, taps of the first call, counters."""
    p = device.default_params(**kw)
    p.flags = flags | device.FLAG_COUNTERS
    ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=calls[0][0], reject_hist=1))
    ctx.reset_counters()
    taps = None
    for i, (first, n) in enumerate(calls):
        if i == 0:
            p.flags |= device.FLAG_TAPS
        else:
            p.flags &= ~device.FLAG_TAPS
        ctx.render(first, n, 1, p)
        if i == 0:
            taps = ctx.read_taps()
    return ctx.read_accum(), taps, ctx.counters()


@pytest.mark.parametrize("sun", [0, 1])
@pytest.mark.parametrize("name", list(SCENES))
def test_short_path_equals_staged_pipeline_bit_for_bit(ctx, scenes, name, sun):
    s, a, res = scenes[name]
    ctx.upload_scene(a)
    kw = dict(max_bounces=5, enable_sun=sun)
    one_by_one = [(1, 1), (2, 1), (3, 1)]
    ref_acc, ref_taps, ref_cnt = render(ctx, s, res, one_by_one, flags=device.FLAG_STAGED, **kw)
    variants = {
        "short, one frame per call": (one_by_one, 0),
        "short, three frames in one call": ([(1, 3)], 0),
        "short, two frames then one": ([(1, 2), (3, 1)], 0),
        "staged, three frames in one call": ([(1, 3)], device.FLAG_STAGED),
    }
    for label, (calls, flags) in variants.items():
        acc, taps, cnt = render(ctx, s, res, calls, flags=flags, **kw)
        assert np.array_equal(acc, ref_acc), label
        for k in ("tri", "prim", "t", "seed"):
            assert np.array_equal(taps[k], ref_taps[k]), (label, k)
        for k in EXACT + ("shadow_rays",):
            assert cnt[k] == ref_cnt[k], (label, k, cnt[k], ref_cnt[k])
    assert np.all(ref_acc[..., 3] == 3.0)
    assert (ref_cnt["shadow_rays"] > 0) == (sun == 1)


@pytest.mark.parametrize("knobs", [dict(RTHIP_PATH_SPLIT="1"), dict(RTHIP_PATH_SPLIT="0"), dict(RTHIP_PATH_LONGEST_FIRST="0"),
                                   dict(RTHIP_PATH_SPLIT="1", RTHIP_SPREAD="2")])
def test_hand_out_order_changes_no_result(ctx, scenes, knobs):
    """The single launch hands its camera paths out in an order made from the costs of the call
    before and from every path's first termination draw (long paths first where a lane gets many
    paths).  On these small images the library would keep every run's paths together: the knobs
    force each form, on images whose edge tiles are partly outside (67 x 41) and with several frames
    per call -- every pixel must get each of its samples exactly once, and the sums must not move."""
    import os
    for name in ("box", "materials"):
        s, a, res = scenes[name]
        ctx.upload_scene(a)
        calls = [(1, 1), (2, 1), (3, 3), (6, 1)]
        ref_acc, ref_taps, ref_cnt = render(ctx, s, res, calls, flags=device.FLAG_STAGED, max_bounces=6)
        old = {k: os.environ.get(k) for k in knobs}
        try:
            os.environ.update(knobs)
            acc, taps, cnt = render(ctx, s, res, calls, max_bounces=6)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert np.all(acc[..., 3] == 6.0), (name, knobs)
        assert np.array_equal(acc, ref_acc), (name, knobs)
        for k in ("tri", "prim", "t", "seed"):
            assert np.array_equal(taps[k], ref_taps[k]), (name, knobs, k)
        for k in EXACT:
            assert cnt[k] == ref_cnt[k], (name, knobs, k, cnt[k], ref_cnt[k])


def test_short_path_is_the_one_that_runs(ctx, scenes):
    """Timing records which launches a call made: a frame group of at most 8 * 2^20 camera paths =
    one k_path launch and no stage launch (15 360 per frame here: any group); more = the staged
    pipeline (a 1080p frame has 2 073 600: four frames run as one launch, more are staged)."""
    s, a, res = scenes["materials"]
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
    p = device.default_params(flags=device.FLAG_TIMING)
    for n in (1, 2, 3, 4, 64, 128):
        ctx.reset_timing()
        ctx.render(1, n, 1, p)
        t = ctx.timing()
        assert t["path"]["launches"] == 1, (n, t)
        assert t["extend"]["launches"] == 0, (n, t)
    ctx.reset_timing()
    ctx.render(1, 129, 1, p)         # two frame groups (128 frames in flight): two launches
    assert ctx.timing()["path"]["launches"] == 2
    ctx.set_frame_uniforms(s.frame_uniforms((1920, 1080), frame=1, reject_hist=1))
    # (the first large call at a resolution renders its first frame as a single launch of its own: the
    # costs of its tiles are what the staged pipeline orders its queue by from then on)
    for n, want_path, want_extend in ((4, 1, 0), (6, 1, 8), (6, 0, 8)):   # eight bounces: eight extend launches
        ctx.reset_timing()
        ctx.render(1, n, 1, p)
        t = ctx.timing()
        assert t["path"]["launches"] == want_path, (n, t)
        assert t["extend"]["launches"] == want_extend, (n, t)
    ctx.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
    # neither a stack deeper than the short path's LDS column nor the sun option asks for the staged pipeline
    for kw, want_path in ((dict(stack_depth=40), 1), (dict(enable_sun=1), 1), (dict(stack_depth=64, enable_sun=1), 1)):
        ctx.reset_timing()
        ctx.render(1, 1, 1, device.default_params(flags=device.FLAG_TIMING, **kw))
        assert ctx.timing()["path"]["launches"] == want_path, kw


@pytest.mark.parametrize("name", ["materials", "atrium"])
def test_short_path_matches_oracle(ctx, scenes, name):
    s, a, res = scenes[name]
    ctx.upload_scene(a)
    acc, taps, cnt = render(ctx, s, res, [(5, 1)], max_bounces=8)
    p = device.default_params(max_bounces=8)
    u = s.frame_uniforms(res, frame=5, reject_hist=1)
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 5, 1, 1, taps=True, n_threads=8)
    n = res[0] * res[1]
    assert float((taps["tri"] == ref["tri"]).mean()) >= 0.9999
    same = taps["tri"] == ref["tri"]
    assert np.array_equal(taps["prim"][same], ref["prim"][same])
    np.testing.assert_allclose(taps["t"][same], ref["t"][same], rtol=1e-5, atol=1e-6)   # the camera ray's sin / cos come from another library
    ok = taps["seed"] == ref["seed"]
    assert float(ok.mean()) >= 0.998
    assert np.array_equal(acc[..., 3], ref["accum"][..., 3])
    np.testing.assert_allclose(acc[ok], ref["accum"][ok], rtol=2e-3, atol=2e-4)
    for k in ("rays", "shade_calls", "misses", "kills", "inner_steps", "tri_tests"):
        assert abs(cnt[k] - ref["counters"][k]) <= max(4, 3e-3 * ref["counters"][k]), k
    assert cnt["paths"] == n


def test_short_path_converges_to_the_oracle_image(ctx, scenes):
    """48 calls of one frame each, the reference's call pattern (main.rs:356-407), against the oracle."""
    s, a, _ = scenes["materials"]
    res = (96, 64)
    ctx.upload_scene(a)
    acc, _, _ = render(ctx, s, res, [(f, 1) for f in range(1, 49)])
    ref = orc.render(oracle_scene(a), s.frame_uniforms(res, frame=1, reject_hist=1), oracle_params(device.default_params()), 1, 48, 1, n_threads=8)
    assert np.all(acc[..., 3] == 48.0)
    assert psnr(tonemapped(acc), tonemapped(ref["accum"])) >= 45.0


@pytest.mark.parametrize("sun", [0, 1])
@pytest.mark.parametrize("depth", [2, 5, 23, 24, 40])
def test_short_path_drops_pushes_like_the_staged_pipeline(ctx, scenes, depth, sun):
    """With the sun option a shadow ray's own dropped push paints the pixel too (shader.wgsl:306-313, 941-943).
    Depths above 23 continue the stack in memory (columns per thread), in both pipelines."""
    s, a, res = scenes["terrain"]
    ctx.upload_scene(a)
    acc0, taps0, cnt0 = render(ctx, s, res, [(1, 2)], flags=device.FLAG_STAGED, stack_depth=depth, enable_sun=sun)
    acc1, taps1, cnt1 = render(ctx, s, res, [(1, 2)], stack_depth=depth, enable_sun=sun)
    assert np.array_equal(acc0, acc1)
    assert np.array_equal(taps0["seed"], taps1["seed"]) and np.array_equal(taps0["tri"], taps1["tri"])
    for k in EXACT:
        assert cnt0[k] == cnt1[k], k
    if depth == 2:
        assert cnt1["stack_overflows"] > 0
        yellow = (acc1[..., 0] == acc1[..., 1]) & (acc1[..., 2] == 0.0) & (acc1[..., 0] > 0)
        assert yellow.any()


def test_short_path_nan_directions(ctx, scenes):
    """Every camera ray with a NaN direction (NaN focus distance): the precomputed walk, as in the staged pipeline."""
    s, a, _ = scenes["materials"]
    res = (16, 8)
    u = np.ascontiguousarray(s.frame_uniforms(res, frame=1, reject_hist=1)).view(np.uint8).reshape(-1).copy()
    u[540:544] = np.array([np.nan], np.float32).view(np.uint8)
    ctx.upload_scene(a)
    out = []
    for flags in (device.FLAG_STAGED, 0):
        p = device.default_params(max_bounces=2, flags=flags | device.FLAG_TAPS | device.FLAG_COUNTERS)
        ctx.set_frame_uniforms(u)
        ctx.reset_counters()
        ctx.render(1, 1, 1, p)
        out.append((ctx.read_accum(), ctx.read_taps(), ctx.counters()))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.all(out[1][1]["tri"] == -1)
    for k in EXACT:
        assert out[0][2][k] == out[1][2][k], k


def test_short_path_into_a_bound_accumulator_and_after_resize(ctx, scenes):
    """The one-frame form adds into `screen` itself: the caller's buffer when one is bound, and the
    new buffer after a change of resolution."""
    import torch
    s, a, _ = scenes["box"]
    ctx.upload_scene(a)
    res = (40, 24)
    t = torch.full((res[0] * res[1] * 4,), 7.0, dtype=torch.float32, device="cuda:0")
    ctx.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))   # the resolution a bound buffer must match
    ctx.bind_accum(t.data_ptr(), res[0] * res[1])
    try:
        ctx.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
        ctx.render(1, 1, 1)
        ctx.render(2, 1, 1)
        ctx.synchronize()
        got = t.cpu().numpy().reshape(res[1], res[0], 4)
    finally:
        ctx.bind_accum(0, 0)
    assert np.all(got[..., 3] == 2.0)
    acc, _, _ = render(ctx, s, res, [(1, 2)], flags=device.FLAG_STAGED)
    assert np.array_equal(got, acc)
    for r in ((64, 64), (24, 40)):
        acc1, _, _ = render(ctx, s, r, [(1, 1), (2, 1)])
        acc0, _, _ = render(ctx, s, r, [(1, 2)], flags=device.FLAG_STAGED)
        assert np.array_equal(acc0, acc1)


def test_staged_pipeline_queues_tiles_by_cost_without_changing_a_bit(ctx, scenes):
    """A group too large for the single launch goes through the staged pipeline with its tiles queued
    expensive first, by the costs a one-frame single launch left behind; a large call that finds none
    for its image renders its first frame that way.  Against RTHIP_FLAG_STAGED (tiles as they lie in
    the image): the same bits, on images whose edge tiles are partly outside, and across changes of
    resolution, after which the order of the old image must not be used."""
    s, a, _ = scenes["atrium"]
    ctx.upload_scene(a)
    frames = 18                                            # 17 x 1003 x 517 > 8 * 2^20 camera paths: staged even after a first frame on its own
    sizes = ((1003, 517), (1100, 480), (1003, 517))
    want = {}
    for res in sizes[:2]:
        want[res] = render(ctx, s, res, [(1, frames)], flags=device.FLAG_STAGED, max_bounces=4)
    for res, again in ((sizes[0], False), (sizes[0], True), (sizes[1], False), (sizes[2], False)):
        ctx.reset_timing()
        acc, taps, cnt = render(ctx, s, res, [(1, frames)], flags=device.FLAG_TIMING, max_bounces=4)
        t = ctx.timing()
        # the first large call at a resolution: one frame as a single launch, the rest staged
        assert t["path"]["launches"] == (0 if again else 1), (res, again, t)
        assert t["extend"]["launches"] == 4, (res, again, t)
        ref_acc, ref_taps, ref_cnt = want[res]
        assert np.all(acc[..., 3] == float(frames))
        assert np.array_equal(acc, ref_acc), (res, again)
        for k in ("tri", "prim", "t", "seed"):
            assert np.array_equal(taps[k], ref_taps[k]), (res, again, k)
        for k in EXACT:
            assert cnt[k] == ref_cnt[k], (res, again, k, cnt[k], ref_cnt[k])
