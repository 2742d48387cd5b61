"""GPU parity at the sizes the benchmark configurations name.

Small cases are compared pixel by pixel in test_gpu_parity.py.  Here the configurations run at
their full resolution: against the oracle where the oracle finishes in seconds (all of the 512x512
cases, sampled rows of the 1080p case), and through properties that do not depend on the size
(bit-identical results for any grouping of the frames, for repeated runs and for striped
accumulation; the sample count in the alpha lane).
"""
import os

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import device, procedural
from util import oracle_params, oracle_scene, psnr, tonemapped

pytestmark = pytest.mark.gpu

THREADS = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def ctx():
    c = device.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def atrium():
    s = procedural.atrium_scene()
    return s, s.arrays()


def render(ctx, scene, arrays, res, first, n, stride=1, params=None, upload=True):
    if upload:
        ctx.upload_scene(arrays)
    ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=first, reject_hist=1))
    ctx.render(first, n, stride, params or device.default_params())
    return ctx.read_accum()


def test_config1_box_512_one_sample(ctx):
    """Box, 512x512, 1 spp, reference parameters (8 bounces, 50 % termination, DoF)."""
    s = procedural.box_scene()
    a = s.arrays()
    res = (512, 512)
    p = device.default_params()
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 1, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 1, 1, 1, taps=True, n_threads=THREADS)
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.9999
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.999
    same = got["seed"] == ref["seed"]
    np.testing.assert_allclose(got["accum"][same], ref["accum"][same], rtol=2e-3, atol=2e-4)
    assert got["counters"]["paths"] == 512 * 512
    assert abs(got["counters"]["rays"] - ref["counters"]["rays"]) <= 2e-3 * ref["counters"]["rays"]


def test_config2_blob_512_64_samples(ctx):
    """Suzanne-sized single primitive (15 744 triangles, metallic 1 / roughness 1), 512x512, 64 spp."""
    s = procedural.blob_scene()
    a = s.arrays()
    res = (512, 512)
    p = device.default_params()
    got = render(ctx, s, a, res, 1, 64, params=p)
    ref = orc.render(oracle_scene(a), s.frame_uniforms(res, frame=1, reject_hist=1), oracle_params(p), 1, 64, 1,
                     n_threads=THREADS)["accum"]
    assert np.all(got[..., 3] == 64.0)
    assert psnr(tonemapped(got), tonemapped(ref)) >= 45.0


def test_config3_atrium_1080p_rows_match_oracle(ctx, atrium):
    """Headline workload at full size: every 135th row of frames 1 and 2 against the oracle."""
    s, a = atrium
    res = (1920, 1080)
    p = device.default_params(max_bounces=4)
    p.flags |= device.FLAG_TAPS
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.render(1, 2, 1, p)
    got = dict(accum=ctx.read_accum(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 1, 2, 1, taps=True, n_threads=THREADS,
                     row_start=67, row_step=135)
    rows = np.arange(67, 1080, 135)
    assert float((got["tri"][rows] == ref["tri"][rows]).mean()) >= 0.9999
    assert float((got["seed"][rows] == ref["seed"][rows]).mean()) >= 0.998
    same = got["seed"][rows] == ref["seed"][rows]
    np.testing.assert_allclose(got["accum"][rows][same], ref["accum"][rows][same], rtol=2e-3, atol=2e-4)
    assert np.all(got["accum"][..., 3] == 2.0)


def test_config3_atrium_1080p_whole_frames_match_oracle(ctx, atrium):
    """Headline workload, every pixel: primary hits and final RNG states of frame 1, and the
    8-sample image against the oracle's (PSNR in display space; pixels whose paths diverged at a
    last-ulp branch are in, not masked out)."""
    s, a = atrium
    res = (1920, 1080)
    p = device.default_params(max_bounces=4)
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 8, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    sc = oracle_scene(a)
    one = orc.render(sc, u, oracle_params(p), 1, 1, 1, taps=True, n_threads=THREADS)
    assert float((got["tri"] == one["tri"]).mean()) >= 0.9999
    hit = (got["tri"] == one["tri"]) & (one["tri"] >= 0)
    assert np.array_equal(got["prim"][hit], one["prim"][hit])
    assert float((got["seed"] == one["seed"]).mean()) >= 0.999
    ref = orc.render(sc, u, oracle_params(p), 1, 8, 1, n_threads=THREADS)
    assert np.all(got["accum"][..., 3] == 8.0) and np.array_equal(ref["accum"][..., 3], got["accum"][..., 3])
    assert psnr(tonemapped(got["accum"]), tonemapped(ref["accum"])) >= 45.0
    for k in ("rays", "misses", "kills", "shade_calls", "alpha_tests"):
        assert abs(got["counters"][k] - ref["counters"][k]) <= 2e-3 * ref["counters"][k], k


def test_config3_frame_grouping_is_invisible(ctx, atrium):
    """72 frames in one call (one group), in groups of 7, and one frame per call give the
    same accumulator bit for bit; so does a second run."""
    s, a = atrium
    res = (1920, 1080)
    p = device.default_params(max_bounces=4)
    n = 72
    whole = render(ctx, s, a, res, 1, n, params=p)
    again = render(ctx, s, a, res, 1, n, params=p, upload=False)
    assert np.array_equal(whole, again)
    old = os.environ.get("RTHIP_FRAMES_IN_FLIGHT")
    try:
        os.environ["RTHIP_FRAMES_IN_FLIGHT"] = "7"
        sevens = render(ctx, s, a, res, 1, n, params=p, upload=False)
    finally:
        if old is None:
            del os.environ["RTHIP_FRAMES_IN_FLIGHT"]
        else:
            os.environ["RTHIP_FRAMES_IN_FLIGHT"] = old
    assert np.array_equal(whole, sevens)
    ctx.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
    for f in range(1, n + 1):
        ctx.render(f, 1, 1, p)
    assert np.array_equal(whole, ctx.read_accum())
    assert np.all(whole[..., 3] == float(n))
    assert np.isfinite(whole).all()


def test_config4_4k_striped_accumulation(ctx, atrium):
    """3840x2160: the frames of two ranks (stride 2) accumulated on top of each other hold the same
    samples as one run over all frames; only the order of the float additions differs."""
    s, a = atrium
    res = (3840, 2160)
    p = device.default_params(max_bounces=4)
    n = 8
    full = render(ctx, s, a, res, 1, n, params=p)
    ctx.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
    ctx.render(1, n // 2, 2, p)     # rank 0 of 2: frames 1, 3, 5, 7
    ctx.render(2, n // 2, 2, p)     # rank 1 of 2: frames 2, 4, 6, 8
    striped = ctx.read_accum()
    assert np.all(full[..., 3] == float(n)) and np.array_equal(striped[..., 3], full[..., 3])
    np.testing.assert_allclose(striped, full, rtol=1e-5, atol=1e-5)
    img = ctx.resolve()
    assert img.shape == (2160, 3840, 4) and np.all(img[..., 3] == 255)


def test_config5_tiled_terrain_deep_stack(ctx):
    """Tiled terrain merged into one BVH (here 12x12 tiles, 1.2 M triangles), stack depth 48: the
    part of the stack beyond the LDS levels lives in global memory."""
    s = procedural.terrain_scene(tiles=12, env_size=(128, 64))
    a = s.arrays()
    res = (256, 144)
    p = device.default_params(max_bounces=4, stack_depth=48)
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 2, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 1, 2, 1, taps=True, n_threads=THREADS)
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.9999
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.999
    same = got["seed"] == ref["seed"]
    np.testing.assert_allclose(got["accum"][same], ref["accum"][same], rtol=2e-3, atol=2e-4)
    assert got["counters"]["stack_overflows"] == ref["counters"]["stack_overflows"] == 0
    # the call above ran as one persistent launch (a small frame group); the staged pipeline gives the same bits
    p.flags |= device.FLAG_STAGED
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 2, 1, p)
    assert np.array_equal(ctx.read_accum(), got["accum"])
    staged = ctx.counters()
    for k in ("rays", "inner_steps", "leaf_steps", "tri_tests", "shade_calls", "misses", "kills"):
        assert staged[k] == got["counters"][k], k


def test_shallow_stack_on_deep_tree_overflows_like_the_oracle(ctx):
    """The same tree with the reference's 23 entries and with 26 (two levels in global memory)."""
    s = procedural.terrain_scene(tiles=6, env_size=(128, 64))
    a = s.arrays()
    res = (128, 72)
    for depth in (4, 26):
        p = device.default_params(max_bounces=2, stack_depth=depth)
        p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
        u = s.frame_uniforms(res, frame=3, reject_hist=1)
        ctx.upload_scene(a)
        ctx.set_frame_uniforms(u)
        ctx.reset_counters()
        ctx.render(3, 1, 1, p)
        got = dict(counters=ctx.counters(), **ctx.read_taps())
        ref = orc.render(oracle_scene(a), u, oracle_params(p), 3, 1, 1, taps=True, n_threads=THREADS)
        assert float((got["tri"] == ref["tri"]).mean()) >= 0.999
        for k in ("inner_steps", "tri_tests", "stack_overflows"):
            assert abs(got["counters"][k] - ref["counters"][k]) <= max(8, 2e-3 * ref["counters"][k]), (depth, k)
        assert (ref["counters"]["stack_overflows"] > 0) == (depth == 4)


def test_many_primitives_span_several_chunks(ctx):
    """288 instanced primitives (three chunks of 128 for the candidate scan, whose chunk and
    group union bounds let whole chunks and groups be skipped): frames and single rays against
    the oracle, which visits every primitive."""
    s = procedural.terrain_scene(tiles=12, env_size=(128, 64), instanced=True)
    a = s.arrays()
    assert a["prims"].shape[0] == 288
    res = (256, 144)
    p = device.default_params(max_bounces=4)
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 2, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    sc = oracle_scene(a)
    ref = orc.render(sc, u, oracle_params(p), 1, 2, 1, taps=True, n_threads=THREADS)
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.9999
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.999
    same = got["seed"] == ref["seed"]
    assert np.array_equal(got["prim"][same & (got["tri"] >= 0)], ref["prim"][same & (ref["tri"] >= 0)])
    np.testing.assert_allclose(got["accum"][same], ref["accum"][same], rtol=2e-3, atol=2e-4)
    assert got["counters"]["prims_visited"] < 0.05 * ref["counters"]["root_tests"]
    # single rays, bit for bit, from points above the terrain in all directions
    rng = np.random.default_rng(3)
    n = 400
    o = np.column_stack([rng.uniform(-230, 230, n), rng.uniform(-230, 230, n), rng.uniform(0.5, 30, n)]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q = ctx.trace_rays(o, d)
    op = oracle_params(device.default_params())
    for i in range(n):
        tri, prim, t, _ = orc.trace_ray(sc, o[i], d[i], op, 0)
        assert q["tri"][i] == tri, i
        if tri >= 0:
            assert q["prim"][i] == prim and q["t"][i] == np.float32(t), i


@pytest.mark.parametrize("tiles,n_prims", [(2, 8), (4, 32), (8, 128), (9, 162)])
def test_primitive_counts_around_the_candidate_mask_limits(ctx, tiles, n_prims):
    """8 and 32 primitives (one mask word, cull stage), 128 (all four words, the last scene the cull
    stage serves) and 162 (two chunks, in-kernel scan): frames against the oracle, which tests every
    primitive."""
    s = procedural.terrain_scene(tiles=tiles, n=16, env_size=(64, 32), instanced=True)
    a = s.arrays()
    assert a["prims"].shape[0] == n_prims
    res = (160, 90)
    p = device.default_params(max_bounces=3)
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=5, reject_hist=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(5, 2, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 5, 2, 1, taps=True, n_threads=THREADS)
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.9995
    hit = (got["tri"] >= 0) & (got["tri"] == ref["tri"])
    assert np.array_equal(got["prim"][hit], ref["prim"][hit])
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.998
    same = got["seed"] == ref["seed"]
    np.testing.assert_allclose(got["accum"][same], ref["accum"][same], rtol=2e-3, atol=2e-4)
    for k in ("inner_steps", "tri_tests"):
        assert abs(got["counters"][k] - ref["counters"][k]) <= max(16, 3e-3 * ref["counters"][k]), k


def test_camera_cull_gives_the_masks_of_the_general_cull(ctx, atrium):
    """The wave-level interval pre-filter of the camera-ray cull must never drop a primitive a ray
    meets: counters (prims_visited counts the mask bits) and results are identical with and
    without it, at full size, over several frames, with depth of field."""
    s, a = atrium
    res = (1920, 1080)
    p = device.default_params(max_bounces=2)
    p.flags |= device.FLAG_COUNTERS | device.FLAG_TAPS
    out = {}
    # general cull / camera cull that also makes the camera rays (the default) / camera cull behind k_generate
    modes = {"0": dict(RTHIP_CULL_CAMERA="0"), "1": dict(RTHIP_CULL_CAMERA="1"), "2": dict(RTHIP_CULL_CAMERA="1", RTHIP_GENERATE_IN_CULL="0")}
    names = ("RTHIP_CULL_CAMERA", "RTHIP_GENERATE_IN_CULL")
    old = {k: os.environ.get(k) for k in names}
    try:
        for mode, env in modes.items():
            for k in names:
                os.environ.pop(k, None)
            os.environ.update(env)
            ctx.upload_scene(a)
            ctx.set_frame_uniforms(s.frame_uniforms(res, frame=3, reject_hist=1))
            ctx.reset_counters()
            ctx.render(3, 6, 1, p)
            out[mode] = (ctx.read_accum(), ctx.counters(), ctx.read_taps())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for mode in ("1", "2"):
        for k in ("rays", "paths", "prims_visited", "inner_steps", "tri_tests", "alpha_tests", "misses"):
            assert out["0"][1][k] == out[mode][1][k], (mode, k)
        assert np.array_equal(out["0"][0], out[mode][0]), mode
        assert np.array_equal(out["0"][2]["tri"], out[mode][2]["tri"]), mode


@pytest.mark.parametrize("walk", ["scan", "pyramid", "list"])
def test_ten_thousand_primitives_use_super_chunks(ctx, monkeypatch, walk):
    """10 368 instanced primitives = 81 chunks of 128, more than the 64 bits of a ray's chunk mask:
    pairs of chunks form super-chunks.  Single rays bit for bit and a frame against the oracle, with
    the chunk scan forced and with the bounds pyramid the library picks for such a list."""
    s = procedural.terrain_scene(tiles=72, n=8, env_size=(64, 32), instanced=True)
    a = s.arrays()
    assert a["prims"].shape[0] == 10368
    if walk == "scan":
        monkeypatch.setenv("RTHIP_TREE_MIN_PRIMS", "4000000000")
    elif walk == "list":   # per-ray candidate lists, for the single rays and the frame below
        monkeypatch.setenv("RTHIP_CANDIDATE_WALK", "list")
    ctx.upload_scene(a)
    assert ctx.candidate_walk() == walk
    sc = oracle_scene(a)
    rng = np.random.default_rng(11)
    n = 300
    o = np.column_stack([rng.uniform(-1400, 1400, n), rng.uniform(-1400, 1400, n), rng.uniform(0.5, 40, n)]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 2] -= 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q = ctx.trace_rays(o, d)
    op = oracle_params(device.default_params())
    hits = 0
    for i in range(n):
        tri, prim, t, _ = orc.trace_ray(sc, o[i], d[i], op, 0)
        assert q["tri"][i] == tri, i
        if tri >= 0:
            hits += 1
            assert q["prim"][i] == prim and q["t"][i] == np.float32(t), i
    assert hits > n // 4
    res = (96, 54)
    p = device.default_params(max_bounces=2)
    p.flags |= device.FLAG_TAPS
    u = s.frame_uniforms(res, frame=1, reject_hist=1)
    ctx.set_frame_uniforms(u)
    ctx.render(1, 1, 1, p)
    got = ctx.read_taps()
    ref = orc.render(sc, u, oracle_params(p), 1, 1, 1, taps=True, n_threads=THREADS)
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.999
    hit = (got["tri"] >= 0) & (got["tri"] == ref["tri"])
    assert np.array_equal(got["prim"][hit], ref["prim"][hit])


@pytest.mark.parametrize("kind", ["atrium_cull_masks", "instanced_in_kernel_scan", "instanced_pyramid", "instanced_pyramid_deep_stack", "instanced_lists", "instanced_lists_deep_stack", "deep_stack"])
def test_sun_shadow_rays_on_every_traversal_variant(ctx, atrium, kind):
    """The sun option's shadow rays (trace_shadow, shader.wgsl:528-575, 630-638: any hit, no alpha
    test, no distance bound) run on the any-hit form of the extend stage: with candidate masks from
    the cull stage (103 primitives, MASK materials that shadow rays must treat as opaque), with the
    in-kernel candidate scan (162 instanced primitives), with the walk of the bounds pyramid and with
    per-ray candidate lists (800 instanced primitives in rows / shuffled) and with a traversal stack
    that continues in global memory."""
    if kind == "atrium_cull_masks":
        s, a = atrium
        res, depth = (320, 180), 23
    elif kind == "instanced_in_kernel_scan":
        s = procedural.terrain_scene(tiles=9, n=16, env_size=(64, 32), instanced=True)
        a, res, depth = s.arrays(), (192, 108), 23
    elif kind.startswith("instanced_pyramid") or kind.startswith("instanced_lists"):
        # 800 primitives, in rows (the pyramid) or shuffled (per-ray lists from the spatial tree)
        s = procedural.terrain_scene(tiles=20, n=12, env_size=(64, 32), instanced=True, shuffled=kind.startswith("instanced_lists"))
        a, res, depth = s.arrays(), (192, 108), (40 if kind.endswith("deep_stack") else 23)
    else:
        s = procedural.terrain_scene(tiles=6, env_size=(64, 32))
        a, res, depth = s.arrays(), (192, 108), 40
    p = device.default_params(max_bounces=4, stack_depth=depth, enable_sun=1)
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=2, reject_hist=1)
    ctx.upload_scene(a)
    assert ctx.candidate_walk() == {"atrium_cull_masks": "mask", "instanced_in_kernel_scan": "scan", "deep_stack": "mask"}.get(kind, "list" if "lists" in kind else "pyramid")
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(2, 3, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 2, 3, 1, taps=True, n_threads=THREADS)
    assert ref["counters"]["shadow_rays"] > 1000
    assert abs(got["counters"]["shadow_rays"] - ref["counters"]["shadow_rays"]) <= max(4, 2e-3 * ref["counters"]["shadow_rays"])
    assert got["counters"]["rays"] == pytest.approx(ref["counters"]["rays"], rel=2e-3)   # shadow rays are not counted as rays
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.997
    same = got["seed"] == ref["seed"]
    bad = ~np.isclose(got["accum"], ref["accum"], rtol=2e-3, atol=2e-4).all(axis=-1) & same
    assert bad.mean() <= 2e-4, int(bad.sum())
    assert np.array_equal(got["accum"][..., 3], ref["accum"][..., 3])


@pytest.mark.timeout(900)
def test_config5_reference_terrain_tiled_to_fifty_million_triangles(ctx):
    """Configuration 5 at its named size: the reference's simple_terrain.glb on a 78 x 78 grid merged
    into one primitive (49.9 M triangles, one BVH from the bit-exact builder), 1920x1080, 16 frames,
    stack depth 48, seen from above so that the image covers the tiling (the HBM-resident case).
    Against the oracle where it is affordable -- sampled rows of frame 1 and single rays, bit for
    bit -- and through size-independent properties: the sample count in the alpha lane, and an
    accumulator that does not depend on how the 16 frames are grouped."""
    from util import tiled_reference_terrain
    s, a = tiled_reference_terrain(78, top_view=True)
    assert a["tris"].shape[0] == 78 * 78 * 8198 and a["prims"].shape[0] == 1
    res = (1920, 1080)
    p = device.default_params(max_bounces=4, stack_depth=48)
    p.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = s.frame_uniforms(res, frame=1, reject_hist=1, camera_idx=1)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 16, 1, p)
    whole = ctx.read_accum()
    taps = ctx.read_taps()
    c = ctx.counters()
    assert np.all(whole[..., 3] == 16.0) and np.isfinite(whole).all()
    assert c["stack_overflows"] == 0 and c["paths"] == 16 * res[0] * res[1]
    assert (taps["tri"] >= 0).mean() > 0.4
    # the same 16 frames in groups of five
    old = os.environ.get("RTHIP_FRAMES_IN_FLIGHT")
    try:
        os.environ["RTHIP_FRAMES_IN_FLIGHT"] = "5"
        ctx.set_frame_uniforms(u)
        ctx.render(1, 16, 1, p)
        assert np.array_equal(whole, ctx.read_accum())
    finally:
        if old is None:
            del os.environ["RTHIP_FRAMES_IN_FLIGHT"]
        else:
            os.environ["RTHIP_FRAMES_IN_FLIGHT"] = old
    # one frame per call on the same tree: one persistent launch with the stack's upper levels in
    # memory, against the staged pipeline, bit for bit
    single = []
    for flags in (0, device.FLAG_STAGED):
        q = device.default_params(max_bounces=4, stack_depth=48, flags=flags)
        ctx.set_frame_uniforms(u)
        ctx.render(1, 1, 1, q)
        ctx.render(2, 1, 1, q)
        single.append(ctx.read_accum())
    assert np.array_equal(single[0], single[1]) and np.all(single[0][..., 3] == 2.0)
    # oracle: every 180th row of frame 1, and 300 single rays over the whole extent
    sc = oracle_scene(a)
    op = oracle_params(p)
    ref = orc.render(sc, u, op, 1, 1, 1, taps=True, n_threads=THREADS, row_start=90, row_step=180)
    rows = np.arange(90, 1080, 180)
    assert float((taps["tri"][rows] == ref["tri"][rows]).mean()) >= 0.9999
    same = taps["tri"][rows] == ref["tri"][rows]
    np.testing.assert_allclose(taps["t"][rows][same], ref["t"][rows][same], rtol=1e-5)
    rng = np.random.default_rng(17)
    v = a["tris"]["v"][::997, :, :3].reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    n = 300
    o = (lo + rng.random((n, 3)) * (hi - lo)).astype(np.float32)
    o[:, 2] = hi[2] + 5.0
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q = ctx.trace_rays(o, d, None, p)
    hits = 0
    for i in range(n):
        tri, prim, t, _ = orc.trace_ray(sc, o[i], d[i], op, 0)
        assert q["tri"][i] == tri, i
        if tri >= 0:
            hits += 1
            assert q["t"][i] == np.float32(t), i
    assert hits > n // 2
