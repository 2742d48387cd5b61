"""State handling of a device context across calls: buffers that follow the resolution, the
reject_hist protocol when a call renders nothing, and the candidate scan at its 64-bit edge."""
import numpy as np
import pytest

from oracle import orc
from raytracer_amd import device, procedural
from util import oracle_params, oracle_scene

pytestmark = pytest.mark.gpu


def force_walk(monkeypatch, walk):
    """Environment that makes the next upload_scene of a scene of more than 128 primitives use `walk`."""
    monkeypatch.delenv("RTHIP_TREE_MIN_PRIMS", raising=False)
    monkeypatch.delenv("RTHIP_CANDIDATE_WALK", raising=False)
    if walk == "pyramid":
        monkeypatch.setenv("RTHIP_TREE_MIN_PRIMS", "8")
    elif walk == "scan":
        monkeypatch.setenv("RTHIP_TREE_MIN_PRIMS", "4000000000")
    elif walk == "list":
        monkeypatch.setenv("RTHIP_CANDIDATE_WALK", "list")


def fresh(scene, arrays, res, first, n, params=None):
    c = device.Context(0)
    try:
        c.upload_scene(arrays)
        c.set_frame_uniforms(scene.frame_uniforms(res, frame=first, reject_hist=1))
        c.render(first, n, 1, params or device.default_params())
        return c.read_accum()
    finally:
        c.close()


@pytest.mark.parametrize("a_res,b_res", [((8, 5), (40, 1)), ((1000, 500), (2000, 250)), ((96, 64), (64, 96))])
def test_resolution_change_at_equal_pixel_count(a_res, b_res):
    """Queue slots go by 8x8 tile: 8x5 has one tile, 40x1 five; 1000x500 has 7875, 2000x250 has
    8000.  A context that rendered the first shape must size its queues again for the second."""
    s = procedural.box_scene(env_size=(64, 32))
    a = s.arrays()
    assert a_res[0] * a_res[1] == b_res[0] * b_res[1]
    c = device.Context(0)
    try:
        c.upload_scene(a)
        for res in (a_res, b_res, a_res):
            c.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=0))  # the change of shape itself resets
            c.render(1, 3, 1, device.default_params())
            got = c.read_accum()
            assert got.shape == (res[1], res[0], 4)
            assert np.array_equal(got, fresh(s, a, res, 1, 3)), res
    finally:
        c.close()


def test_larger_frame_group_keeps_the_image():
    """Growing the work buffers (a later call with more frames in flight) must not lose what the
    library-owned accumulator already holds."""
    s = procedural.box_scene(env_size=(64, 32))
    a = s.arrays()
    res = (64, 48)
    c = device.Context(0)
    try:
        c.upload_scene(a)
        c.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
        c.render(1, 2, 1, device.default_params())
        c.render(3, 9, 1, device.default_params())
        assert np.array_equal(c.read_accum(), fresh(s, a, res, 1, 11))
    finally:
        c.close()


def test_render_of_zero_frames_settles_reject_hist():
    """A GPU of a group that gets no frame of a short render: its accumulator holds zero samples
    afterwards, and a sum that a reduce then leaves there is added to, not overwritten."""
    import torch
    s = procedural.box_scene(env_size=(64, 32))
    a = s.arrays()
    res = (48, 32)
    n = res[0] * res[1]
    one = fresh(s, a, res, 7, 1)
    c = device.Context(0)
    try:
        acc = torch.full((n * 4,), 3.0, dtype=torch.float32, device="cuda:0")
        c.upload_scene(a)
        c.bind_accum(acc.data_ptr(), n)
        c.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
        c.render(1, 0, 2, device.default_params())
        c.synchronize()
        assert not acc.cpu().numpy().any()
        acc.fill_(5.0)  # what a reduce to this GPU would have left
        torch.cuda.synchronize()
        c.set_frame_uniforms(s.frame_uniforms(res, frame=7, reject_hist=0))
        c.render(7, 1, 1, device.default_params())
        c.synchronize()
        got = acc.cpu().numpy().reshape(res[1], res[0], 4)
        assert np.array_equal(got, one + np.float32(5.0))
        c.bind_accum(0, 0)
    finally:
        c.close()


@pytest.mark.parametrize("walk", ["scan", "pyramid"])
def test_sixty_four_super_chunks_and_rays_that_meet_none(monkeypatch, walk):
    """8192 instanced primitives = 64 chunks of 128 = 64 super-chunks: every bit of a ray's chunk
    mask names a real chunk, so a ray that meets none has an empty mask.  In the pyramid walk such a
    ray meets no entity of the top record."""
    s = procedural.terrain_scene(tiles=64, n=8, env_size=(64, 32), instanced=True)
    a = s.arrays()
    assert a["prims"].shape[0] == 8192
    if walk == "scan":
        monkeypatch.setenv("RTHIP_TREE_MIN_PRIMS", "4000000000")
    c = device.Context(0)
    try:
        c.upload_scene(a)
        assert c.candidate_walk() == walk
        sc = oracle_scene(a)
        rng = np.random.default_rng(5)
        n = 256
        o = np.column_stack([rng.uniform(-1200, 1200, n), rng.uniform(-1200, 1200, n), rng.uniform(0.5, 40, n)]).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[:, 2] -= 0.3
        # a quarter of the rays start far above everything and point up: they meet no bound at all
        o[::4, 2] = 5000.0
        d[::4, 2] = np.abs(d[::4, 2]) + 0.5
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        q = c.trace_rays(o, d)
        op = oracle_params(device.default_params())
        hits = 0
        for i in range(n):
            tri, prim, t, _ = orc.trace_ray(sc, o[i], d[i], op, 0)
            assert q["tri"][i] == tri, i
            if tri >= 0:
                hits += 1
                assert q["prim"][i] == prim and q["t"][i] == np.float32(t), i
        assert np.all(q["tri"][::4] == -1) and np.all(q["t"][::4] == np.float32(99999999.0))
        assert hits > n // 8
    finally:
        c.close()


def test_frames_in_flight_reports_the_group_size_in_use(monkeypatch):
    s = procedural.box_scene(env_size=(64, 32))
    c = device.Context(0)
    try:
        c.upload_scene(s.arrays())
        with pytest.raises(device.RtHipError, match="set_frame_uniforms first"):
            c.frames_in_flight()
        c.set_frame_uniforms(s.frame_uniforms((1920, 1080), frame=1, reject_hist=1))
        assert c.frames_in_flight() == 128
        c.set_frame_uniforms(s.frame_uniforms((3840, 2160), frame=1, reject_hist=1))
        assert 64 <= c.frames_in_flight() < 128         # 96 GiB of work buffers at most (and half of the free memory)
        c.set_frame_uniforms(s.frame_uniforms((7680, 4320), frame=1, reject_hist=1))
        assert 8 <= c.frames_in_flight() < 32
        monkeypatch.setenv("RTHIP_FRAMES_IN_FLIGHT", "7")
        assert c.frames_in_flight() == 7
    finally:
        c.close()


@pytest.mark.parametrize("tiles,n_mesh", [(9, 8), (16, 8), (33, 4)])
def test_bounds_pyramid_walk_equals_chunk_scan(monkeypatch, tiles, n_mesh):
    """Scenes of many primitives take their candidates from a walk of an 8-ary pyramid of union
    bounds, or from per-ray lists made by a walk of a spatial tree, instead of the chunk scan.  The
    ground camera looks along the rows of tiles, so many rays have more candidates than a list
    holds and finish with the scan.  Forced on for small scenes (162, 512 and 2178 primitives: three and
    four levels, full and ragged last records), it must give the same image, taps and traversal counters as
    the scan, and ray queries equal to the oracle's."""
    s = procedural.terrain_scene(tiles=tiles, n=n_mesh, env_size=(64, 32), instanced=True)
    a = s.arrays()
    assert a["prims"].shape[0] == 2 * tiles * tiles
    res = (160, 96)
    p = device.default_params()
    p.flags |= device.FLAG_COUNTERS | device.FLAG_TAPS

    def run(walk):
        force_walk(monkeypatch, walk)
        c = device.Context(0)
        try:
            c.upload_scene(a)
            assert c.candidate_walk() == walk
            c.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
            c.reset_counters()
            c.render(1, 3, 1, p)
            out = dict(accum=c.read_accum(), counters=c.counters(), **c.read_taps())
            rng = np.random.default_rng(11)
            n = 192
            ext = 20.0 * tiles
            o = np.column_stack([rng.uniform(-ext, ext, n), rng.uniform(-ext, ext, n), rng.uniform(0.5, 30, n)]).astype(np.float32)
            d = rng.normal(size=(n, 3)).astype(np.float32)
            d[:, 2] -= 0.3
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            out["q"] = c.trace_rays(o, d)
            out["rays"] = (o, d)
            return out
        finally:
            c.close()

    scan, tree, lists = run("scan"), run("pyramid"), run("list")
    for other in (tree, lists):
        assert np.array_equal(scan["accum"], other["accum"])
        for k in ("tri", "prim", "t", "seed"):
            assert np.array_equal(scan[k], other[k]), k
        for k in ("rays", "prims_visited", "inner_steps", "leaf_steps", "tri_tests", "shade_calls", "misses", "kills", "stack_overflows", "paths"):
            assert scan["counters"][k] == other["counters"][k], k
    sc = oracle_scene(a)
    op = oracle_params(device.default_params())
    o, d = tree["rays"]
    hits = 0
    for i in range(len(o)):
        tri, prim, t, _ = orc.trace_ray(sc, o[i], d[i], op, 0)
        assert tree["q"]["tri"][i] == tri, i
        if tri >= 0:
            hits += 1
            assert tree["q"]["prim"][i] == prim and tree["q"]["t"][i] == np.float32(t), i
    assert hits > len(o) // 8


def test_pyramid_for_coherent_and_lists_for_incoherent_long_primitive_lists(monkeypatch):
    """Row-major tiles: consecutive primitives are neighbours, bounds over index ranges are tight.
    The same instances in shuffled order: every range spans the scene, and the candidates come from
    per-ray lists made with a spatial tree instead.  Either way the image is the chunk scan's (the
    order of the list differs between the two scenes, so each is compared with itself)."""
    monkeypatch.delenv("RTHIP_TREE_MIN_PRIMS", raising=False)
    kinds = {}
    for shuffled in (False, True):
        s = procedural.terrain_scene(tiles=32, n=4, env_size=(64, 32), instanced=True, shuffled=shuffled)
        a = s.arrays()
        assert a["prims"].shape[0] == 2048
        c = device.Context(0)
        try:
            c.upload_scene(a)
            kinds[shuffled] = c.candidate_walk()
            c.set_frame_uniforms(s.frame_uniforms((96, 64), frame=1, reject_hist=1))
            c.render(1, 2, 1, device.default_params())
            own = c.read_accum()
            monkeypatch.setenv("RTHIP_TREE_MIN_PRIMS", "4000000000")
            c.upload_scene(a)
            assert c.candidate_walk() == "scan"
            c.set_frame_uniforms(s.frame_uniforms((96, 64), frame=1, reject_hist=1))
            c.render(1, 2, 1, device.default_params())
            assert np.array_equal(own, c.read_accum())
            monkeypatch.delenv("RTHIP_TREE_MIN_PRIMS")
        finally:
            c.close()
    assert kinds == {False: "pyramid", True: "list"}
    small = procedural.box_scene(env_size=(64, 32))
    c = device.Context(0)
    try:
        with pytest.raises(device.RtHipError, match="no scene"):
            c.candidate_walk()
        c.upload_scene(small.arrays())
        assert c.candidate_walk() == "mask"
    finally:
        c.close()


def test_pyramid_walk_keeps_the_order_that_blend_alpha_draws_depend_on(monkeypatch):
    """BLEND materials draw from the path's random stream inside the traversal (shader.wgsl:464-487),
    once per candidate hit, so the image depends on the order in which the primitives are visited.
    290 MASK and BLEND cards over a floor, in rows: the pyramid walk must give the scan's image, taps
    and alpha-test counts bit for bit, and the oracle's random-stream states."""
    from raytracer_amd import layouts as L
    from raytracer_amd import host
    from oracle import orc as _orc
    s = host.Scene()
    leaves = s.add_texture(procedural.texture_leaves(64, 2))
    floor = procedural.grid(16, 16, (40.0, 40.0), uv_scale=6.0)
    s.add_primitive(floor[0], floor[1], normals=floor[2], uv0=floor[3], material=procedural.material(albedo=(0.7, 0.7, 0.7, 1)))
    card = procedural.grid(3, 3, (1.6, 1.6), uv_scale=1.0)
    where = lambda i, j: host.trs((-16.0 + 2.6 * i, -14.0 + 2.4 * j, 1.0 + 0.1 * ((i + j) % 3)), rot_y=np.pi / 2, rot_z=0.2 * j)
    base = [s.add_primitive(card[0], card[1], normals=card[2], uv0=card[3], transform=where(0, k),
                            material=procedural.material(albedo_tex=leaves, alpha_mode=mode, alpha_cutoff=0.4, roughness=0.8))
            for k, mode in enumerate((L.ALPHA_MASK, L.ALPHA_BLEND))]
    for i in range(12):
        for j in range(12):
            if i or j > 1:
                s.add_instance(base[(i + j) & 1], where(i, j))
    s.set_env(procedural.sky_env(64, 32))
    s.add_camera(host.Camera.look((-24.0, -1.0, 2.2), yaw=0.0, pitch=np.radians(-4), fovy=np.radians(50), focus=10.0))
    a = s.arrays()
    assert a["prims"].shape[0] == 145
    res = (320, 160)
    p = device.default_params(max_bounces=4)
    p.flags |= device.FLAG_COUNTERS | device.FLAG_TAPS
    u = s.frame_uniforms(res, frame=3, reject_hist=1)
    out = {}
    for walk in ("scan", "pyramid", "list"):
        force_walk(monkeypatch, walk)
        c = device.Context(0)
        try:
            c.upload_scene(a)
            assert c.candidate_walk() == walk
            c.set_frame_uniforms(u)
            c.reset_counters()
            c.render(3, 2, 1, p)
            out[walk] = dict(accum=c.read_accum(), counters=c.counters(), **c.read_taps())
        finally:
            c.close()
    assert out["scan"]["counters"]["alpha_tests"] > 20000
    for walk in ("pyramid", "list"):
        assert np.array_equal(out["scan"]["accum"], out[walk]["accum"]), walk
        for k in ("tri", "prim", "t", "seed"):
            assert np.array_equal(out["scan"][k], out[walk][k]), (walk, k)
        for k in ("rays", "prims_visited", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests", "texel_fetches", "misses", "kills"):
            assert out["scan"]["counters"][k] == out[walk]["counters"][k], (walk, k)
    ref = _orc.render(oracle_scene(a), u, oracle_params(p), 3, 2, 1, taps=True, n_threads=16)
    got = out["pyramid"]
    assert float((got["tri"] == ref["tri"]).mean()) >= 0.999
    assert float((got["seed"] == ref["seed"]).mean()) >= 0.997
    assert abs(got["counters"]["alpha_tests"] - ref["counters"]["alpha_tests"]) <= 2e-3 * ref["counters"]["alpha_tests"]


@pytest.mark.parametrize("walk", ["scan", "pyramid"])
def test_coincident_instances_resolve_ties_by_primitive_order(monkeypatch, walk):
    """Every tile placed twice with the same transform: both copies give the same t, and the
    reference keeps the first one found (a later primitive wins only if strictly closer,
    shader.wgsl:607-614).  The copies lie 200 places apart in the primitive list, in different
    chunks and different records of the pyramid."""
    from raytracer_amd import host
    size = 40.0
    t = procedural.terrain_mesh(8, size)
    s = host.Scene()
    mat = procedural.material(albedo=(0.8, 0.8, 0.8, 1))
    pos = lambda k: host.trs(((k % 14 - 7 + 0.5) * size, (k // 14 - 7 + 0.5) * size, 0.0))
    first = s.add_primitive(t[0], t[1], normals=t[2], uv0=t[3], material=mat, transform=pos(0))
    for rep in range(2):
        for k in range(200):
            if rep or k:
                s.add_instance(first, pos(k))
    s.set_env(procedural.sky_env(64, 32))
    s.add_camera(host.Camera.look((0.0, 0.0, 30.0), yaw=0.0, pitch=np.radians(-60), fovy=np.radians(60), focus=30.0))
    a = s.arrays()
    assert a["prims"].shape[0] == 400
    monkeypatch.setenv("RTHIP_TREE_MIN_PRIMS", "8" if walk == "pyramid" else "4000000000")
    c = device.Context(0)
    try:
        c.upload_scene(a)
        assert c.candidate_walk() == walk
        rng = np.random.default_rng(3)
        n = 400
        o = np.column_stack([rng.uniform(-270, 270, n), rng.uniform(-270, 270, n), rng.uniform(5, 40, n)]).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[:, 2] = -np.abs(d[:, 2]) - 0.2
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        q = c.trace_rays(o, d)
        sc = oracle_scene(a)
        op = oracle_params(device.default_params())
        hits = 0
        for i in range(n):
            tri, prim, tt, _ = orc.trace_ray(sc, o[i], d[i], op, 0)
            assert q["tri"][i] == tri and (tri < 0 or (q["prim"][i] == prim and q["t"][i] == np.float32(tt))), i
            hits += tri >= 0
        assert hits > n // 2
        assert np.all(q["prim"][q["tri"] >= 0] < 200)      # always the first copy
    finally:
        c.close()


def test_resize_without_a_render_gives_an_empty_image_not_the_old_buffer():
    """update_resolution (main.rs:79-94) replaces `screen` at once.  After set_frame_uniforms with a
    LARGER resolution and before any render, read_accum and resolve must see a zeroed accumulator of
    the new size, never the old, smaller allocation."""
    s = procedural.box_scene(env_size=(64, 32))
    a = s.arrays()
    c = device.Context(0)
    try:
        c.upload_scene(a)
        c.set_frame_uniforms(s.frame_uniforms((32, 16), frame=1, reject_hist=1))
        c.render(1, 2, 1, device.default_params())
        assert np.all(c.read_accum()[..., 3] == 2.0)
        c.set_frame_uniforms(s.frame_uniforms((640, 480), frame=1, reject_hist=0))
        img = c.resolve()
        assert img.shape == (480, 640, 4) and np.all(img[..., :3] == 0) and np.all(img[..., 3] == 255)
        acc = c.read_accum()
        assert acc.shape == (480, 640, 4) and not acc.any()
        c.render(1, 1, 1, device.default_params())
        assert np.all(c.read_accum()[..., 3] == 1.0)
        # and back down: again the new size, empty
        c.set_frame_uniforms(s.frame_uniforms((24, 24), frame=1, reject_hist=0))
        assert not c.read_accum().any()
    finally:
        c.close()


def test_resolve_into_device_or_pinned_memory_equals_resolve_to_host():
    """rthip_resolve writes the image in place when it is handed memory of its own GPU (the
    presentation buffer of the reference's render pass, main.rs:395-401, lives on the device),
    copies it to pinned host memory like to any other, and allocates nothing in the first case."""
    import torch
    s = procedural.material_test_scene(tex=32, env_size=(64, 32))
    c = device.Context(0)
    try:
        c.upload_scene(s.arrays())
        res = (97, 41)
        c.set_frame_uniforms(s.frame_uniforms(res, frame=1, reject_hist=1))
        c.render(1, 4, 1, device.default_params())
        on_dev = torch.zeros(res[0] * res[1] * 4 + 8, dtype=torch.uint8, device="cuda:0")
        pinned = torch.zeros(res[0] * res[1] * 4, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        c.resolve_into(on_dev.data_ptr())          # before any resolve to the host: no staging image exists yet
        assert torch.cuda.mem_get_info()[0] == free0
        want = c.resolve()
        got = on_dev.cpu().numpy()
        assert np.array_equal(got[:-8].reshape(res[1], res[0], 4), want) and not got[-8:].any()
        c.resolve_into(pinned.data_ptr())
        assert np.array_equal(pinned.numpy().reshape(res[1], res[0], 4), want)
        c.resolve_into(on_dev.data_ptr(), apply_srgb_oetf=False)
        assert np.array_equal(on_dev.cpu().numpy()[:-8].reshape(res[1], res[0], 4), c.resolve(apply_srgb_oetf=False))
    finally:
        c.close()
