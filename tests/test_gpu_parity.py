"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars (north_star): BVH node buffer and integer RNG streams bit-exact; primary-hit triangle ids
equal on >= 99.99 % of pixels; images >= 45 dB PSNR.  Floating-point tolerance exists only
because the device's sin/cos/atan2/asin/pow differ from the C library's in the last ulp.
"""
import numpy as np
import pytest

from oracle import orc
from raytracer_amd import device, procedural
from util import oracle_params, oracle_scene, psnr, tonemapped

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "staged"])
def pipeline(request, monkeypatch):
    """Every test of this module runs twice: as the library chooses (small frame groups = one
    persistent launch, k_path) and with every group forced through the staged pipeline."""
    if request.param == "staged":
        monkeypatch.setenv("RTHIP_FORCE_STAGED", "1")
    return request.param

SCENES = {
    "box": (procedural.box_scene, dict(env_size=(128, 64))),
    "blob": (procedural.blob_scene, dict(env_size=(256, 128))),
    "materials": (procedural.material_test_scene, dict(tex=64, env_size=(128, 64))),
    "terrain": (procedural.terrain_scene, dict(env_size=(128, 64))),
}


@pytest.fixture(scope="module")
def ctx():
    c = device.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def scenes():
    out = {}
    for name, (fn, kw) in SCENES.items():
        s = fn(**kw)
        out[name] = (s, s.arrays())
    return out


def run_both(ctx, scene, arrays, res, n_frames=1, first_frame=1, stride=1, params=None, threads=8):
    params = params or device.default_params()
    params.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = scene.frame_uniforms(res, frame=first_frame, reject_hist=1)
    ctx.upload_scene(arrays)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(first_frame, n_frames, stride, params)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(arrays), u, oracle_params(params), first_frame, n_frames, stride,
                     taps=True, n_threads=threads)
    return got, ref


@pytest.mark.parametrize("name", list(SCENES))
def test_device_tree_is_the_uploaded_tree(ctx, scenes, name):
    _, a = scenes[name]
    ctx.upload_scene(a)
    dev = ctx.read_nodes_device_layout()
    imap = ctx.read_node_index_map()
    host_nodes = a["nodes"]
    back = dev[imap].copy()
    inner = back["count"] == 0
    # undo the index shift of inner nodes through the inverse map
    inv = np.full(dev.shape[0], -1, dtype=np.int64)
    inv[imap] = np.arange(imap.size)
    back["first"][inner] = inv[back["first"][inner]]
    assert back.tobytes() == host_nodes.tobytes()
    # sibling pairs start on 64-byte lines
    assert np.all(dev["first"][imap][inner] % 2 == 0)
    # order-preserving triangle map and (p0, e1, e2) records
    order = ctx.read_tri_order()
    assert np.array_equal(order, np.arange(order.size))
    t = ctx.read_tris_device_layout()
    v = a["tris"]["v"]
    assert np.array_equal(t[:, 0, :3], v[:, 0, :3])
    assert np.array_equal(t[:, 1, :3], v[:, 1, :3] - v[:, 0, :3])
    assert np.array_equal(t[:, 2, :3], v[:, 2, :3] - v[:, 0, :3])


@pytest.mark.parametrize("name,res", [("box", (64, 64)), ("blob", (128, 128)), ("materials", (160, 96)),
                                       ("terrain", (128, 64)), ("materials", (37, 21)), ("blob", (1, 1))])
def test_one_frame_matches_oracle(ctx, scenes, name, res):
    s, a = scenes[name]
    got, ref = run_both(ctx, s, a, res)
    n = res[0] * res[1]
    hit_agree = float((got["tri"] == ref["tri"]).mean())
    assert hit_agree >= (0.9999 if n >= 10000 else 1.0 - 2.0 / n - 1e-9), hit_agree
    same = got["tri"] == ref["tri"]
    assert np.array_equal(got["prim"][same], ref["prim"][same])
    np.testing.assert_allclose(got["t"][same], ref["t"][same], rtol=1e-5, atol=1e-6)
    # integer RNG: a pixel whose path took the same branches ends in the same state
    seed_agree = float((got["seed"] == ref["seed"]).mean())
    assert seed_agree >= (0.999 if n >= 10000 else 1.0 - 2.0 / n - 1e-9), seed_agree
    assert np.array_equal(got["accum"][..., 3], ref["accum"][..., 3])
    ok = got["seed"] == ref["seed"]
    np.testing.assert_allclose(got["accum"][ok], ref["accum"][ok], rtol=2e-3, atol=2e-4)
    for k in ("rays", "shade_calls", "misses", "kills"):
        assert abs(got["counters"][k] - ref["counters"][k]) <= max(4, 2e-3 * ref["counters"][k]), k
    assert got["counters"]["paths"] == n


def test_converged_image_psnr(ctx, scenes):
    s, a = scenes["blob"]
    got, ref = run_both(ctx, s, a, (256, 256), n_frames=32)
    p = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert p >= 45.0, p
    assert np.array_equal(got["accum"][..., 3], np.full((256, 256), 32.0, np.float32))


def test_textured_image_psnr(ctx, scenes):
    s, a = scenes["materials"]
    got, ref = run_both(ctx, s, a, (192, 128), n_frames=32)
    p = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert p >= 45.0, p


def test_resolve_matches_oracle(ctx, scenes):
    s, a = scenes["materials"]
    got, _ = run_both(ctx, s, a, (96, 64), n_frames=8)
    for srgb in (True, False):
        img = ctx.resolve(apply_srgb_oetf=srgb)
        ref = orc.resolve(got["accum"], apply_srgb_oetf=srgb)
        assert np.array_equal(img[..., 3], ref[..., 3])
        d = np.abs(img.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_accumulation_protocol(ctx, scenes):
    """reject_hist = 1 overwrites, later frames add, frame striping covers the same samples."""
    s, a = scenes["box"]
    res = (48, 32)
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(s.frame_uniforms(res, reject_hist=1))
    ctx.render(1, 6, 1)
    full = ctx.read_accum()
    ctx.set_frame_uniforms(s.frame_uniforms(res, reject_hist=1))
    ctx.render(1, 3, 2)   # frames 1, 3, 5
    ctx.render(2, 3, 2)   # frames 2, 4, 6 add on top
    striped = ctx.read_accum()
    assert np.array_equal(striped[..., 3], full[..., 3])
    np.testing.assert_allclose(striped, full, rtol=1e-5, atol=1e-6)
    # a new reject_hist = 1 discards history
    ctx.set_frame_uniforms(s.frame_uniforms(res, reject_hist=1))
    ctx.render(7, 1, 1)
    assert np.all(ctx.read_accum()[..., 3] == 1.0)
    # write_accum restores a checkpoint and rendering continues on top of it
    ctx.write_accum(full)
    ctx.render(7, 1, 1)
    assert np.all(ctx.read_accum()[..., 3] == 7.0)


@pytest.mark.parametrize("kw", [dict(max_bounces=1), dict(max_bounces=4), dict(term_prob=0.0, max_bounces=3),
                                dict(term_prob=0.25), dict(aperture_radius=0.0), dict(stack_depth=3),
                                dict(stack_depth=40)])
def test_parameter_variants_match_oracle(ctx, scenes, kw):
    s, a = scenes["materials"]
    got, ref = run_both(ctx, s, a, (96, 64), n_frames=2, params=device.default_params(**kw))
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.9995
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.995
    assert psnr(tonemapped(got["accum"]), tonemapped(ref["accum"])) >= 40.0
    if kw.get("stack_depth") == 3:   # dropped pushes paint the pixel (1, 1, 0): shader.wgsl:941-943
        assert ref["counters"]["stack_overflows"] > 0
        assert got["counters"]["stack_overflows"] > 0


@pytest.mark.parametrize("depth", [23, 3])
def test_nan_direction_rays_match_oracle(ctx, scenes, depth):
    """A NaN focus distance makes every camera ray's direction NaN.  The reference then fails no
    box test and passes no triangle test: every node of every primitive is visited and the ray
    misses.  The device takes that outcome from a host-side walk of the tree; hits, dropped pushes
    and every step counter must equal the oracle's, which walks the tree ray by ray."""
    s, a = scenes["materials"]
    res = (16, 8)
    params = device.default_params(max_bounces=2, stack_depth=depth)
    params.flags |= device.FLAG_TAPS | device.FLAG_COUNTERS
    u = np.ascontiguousarray(s.frame_uniforms(res, frame=1, reject_hist=1)).view(np.uint8).reshape(-1).copy()
    u[540:544] = np.array([np.nan], np.float32).view(np.uint8)   # scene.camera.focus, shader.wgsl:769
    ctx.upload_scene(a)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(1, 2, 1, params)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(params), 1, 2, 1, taps=True, n_threads=4)
    assert np.all(ref["tri"] == -1) and np.array_equal(got["tri"], ref["tri"])
    assert np.array_equal(got["seed"], ref["seed"])
    for k in ("rays", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests", "stack_overflows", "misses"):
        assert got["counters"][k] == ref["counters"][k], k
    assert got["counters"]["prims_visited"] == ref["counters"]["root_tests"]   # no primitive is culled
    assert (ref["counters"]["stack_overflows"] > 0) == (depth == 3)
    np.testing.assert_array_equal(got["accum"], ref["accum"])


@pytest.mark.parametrize("name,res", [("materials", (160, 96)), ("blob", (96, 96))])
def test_sun_option_matches_oracle(ctx, scenes, name, res):
    """enable_sun = 1: the block the reference keeps commented out (shader.wgsl:916-937); a
    terminated path sends one shadow ray (trace_shadow, :528-575) to the sun."""
    s, a = scenes[name]
    p = device.default_params(max_bounces=4)
    p.enable_sun = 1
    got, ref = run_both(ctx, s, a, res, n_frames=4, params=p)
    assert ref["counters"]["shadow_rays"] > 0
    assert abs(got["counters"]["shadow_rays"] - ref["counters"]["shadow_rays"]) <= max(4, 2e-3 * ref["counters"]["shadow_rays"])
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.9995
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.995
    assert np.array_equal(got["accum"][..., 3], ref["accum"][..., 3])
    assert psnr(tonemapped(got["accum"]), tonemapped(ref["accum"])) >= 45.0
    # the sun must actually light something
    p.enable_sun = 0
    dark, _ = run_both(ctx, s, a, res, n_frames=4, params=p)
    assert float(got["accum"][..., :3].sum()) > float(dark["accum"][..., :3].sum()) * 1.02


def test_sun_shadow_ray_with_a_dropped_push_paints_the_pixel(ctx, scenes):
    """trace_bvh_shadow pushes through the same 23-entry stack (shader.wgsl:528-575): a push it has to
    drop sets `debug` like any other (306-313), and the pixel ends yellow (941-943), also when the
    camera path's own traversal dropped nothing.  A stack of one entry makes that common."""
    s, a = scenes["terrain"]
    p = device.default_params(max_bounces=2, stack_depth=1)
    p.enable_sun = 1
    got, ref = run_both(ctx, s, a, (96, 64), n_frames=1, params=p)
    same = got["seed"] == ref["seed"]
    assert float(same.mean()) >= 0.995
    yellow_ref = (ref["accum"][..., 0] == 1.0) & (ref["accum"][..., 1] == 1.0) & (ref["accum"][..., 2] == 0.0)
    yellow_got = (got["accum"][..., 0] == 1.0) & (got["accum"][..., 1] == 1.0) & (got["accum"][..., 2] == 0.0)
    assert yellow_ref.any()
    assert np.array_equal(yellow_got[same], yellow_ref[same])
    np.testing.assert_allclose(got["accum"][same], ref["accum"][same], rtol=2e-3, atol=2e-4)
    # some of those pixels are yellow ONLY because of the shadow ray: without the sun they are not
    p.enable_sun = 0
    _, ref_dark = run_both(ctx, s, a, (96, 64), n_frames=1, params=p)
    yellow_dark = (ref_dark["accum"][..., 0] == 1.0) & (ref_dark["accum"][..., 1] == 1.0) & (ref_dark["accum"][..., 2] == 0.0)
    assert (yellow_ref & ~yellow_dark).any()


def test_errors_are_reported_not_thrown(scenes):
    c = device.Context(0)
    with pytest.raises(device.RtHipError) as e:
        c.render()
    assert e.value.code == -3
    _, a = scenes["box"]
    bad = dict(a)
    nodes = a["nodes"].copy()
    nodes["first"][0] = 10 ** 6
    bad["nodes"] = nodes
    with pytest.raises(device.RtHipError) as e:
        c.upload_scene(bad)
    assert e.value.code == -1
    with pytest.raises(device.RtHipError):
        device.Context(99)
    c.close()


def test_ray_queries_match_oracle_bit_for_bit(ctx, scenes):
    """rthip_trace_rays against the oracle's trace(), ray by ray: random rays, axis-parallel
    directions (zero components), unnormalised and tiny directions, origins far outside, NaN
    directions, and a stack too short for the tree."""
    s, a = scenes["materials"]
    ctx.upload_scene(a)
    sc = oracle_scene(a)
    rng = np.random.default_rng(7)
    n = 600
    o = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:60] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 60)] * rng.choice([-1.0, 1.0], (60, 1)).astype(np.float32)
    d[60:90] *= 7.5                       # not normalised
    d[90:110] *= 1e-20                    # nearly degenerate
    o[110:130] = o[110:130] * 1e4         # far away
    d[110:130] = -o[110:130] / np.linalg.norm(o[110:130], axis=1, keepdims=True)   # aimed at the scene
    d[130:134] = np.nan
    d[134] = (0.0, 0.0, 0.0)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    for depth in (23, 2, 40):
        p = device.default_params(stack_depth=depth)
        got = ctx.trace_rays(o, d, seeds, p)
        op = oracle_params(p)
        for i in range(n):
            tri, prim, t, seed = orc.trace_ray(sc, o[i], d[i], op, int(seeds[i]))
            assert got["tri"][i] == tri, (depth, i)
            if tri >= 0:
                assert got["prim"][i] == prim, (depth, i)
                assert got["t"][i] == np.float32(t), (depth, i)
            assert got["seed"][i] == seed, (depth, i)
