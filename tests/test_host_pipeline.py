"""The host-side scene pipeline (librthost.so) against the oracle's restatement of scene.rs and
against closed forms: BVH builder bit-exactness, record packing, cameras, frame uniforms."""
import os

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import host, layouts as L, procedural

import gltf_mini


def random_tris(n, seed, spread=10.0, size=0.5):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3))
    p = (c + rng.normal(0, size, (n, 3, 3))).astype(np.float32)
    out = np.zeros(n, dtype=L.TRI)
    out["v"][:, :, :3] = p
    out["v"][:, :, 3] = ((p[:, 0] + p[:, 1]) + p[:, 2]) / np.float32(3.0)
    return out


def assert_same_tree(tris, offset, size, n_threads=0):
    n_o, i_o = orc.bvh_build(tris, offset, size)
    n_p, i_p = host.bvh_build(tris, offset, size, n_threads=n_threads)
    assert n_p.view(np.uint8).tobytes() == n_o.tobytes()
    assert np.array_equal(i_o, i_p)
    return n_p


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 63, 64, 65, 200, 5000])
def test_builder_bit_exact_random(n):
    assert_same_tree(random_tris(n, n), 0, n)


def test_builder_bit_exact_with_offset_and_threads():
    tris = random_tris(30000, 5)
    for off, size in [(0, 30000), (1000, 20000), (29990, 10)]:
        a = assert_same_tree(tris, off, size, n_threads=1)
        b = assert_same_tree(tris, off, size, n_threads=8)
        assert a.tobytes() == b.tobytes()


def test_builder_degenerate_inputs():
    # identical centroids cannot be split: the node stays a leaf with > 2 triangles (scene.rs:1030)
    t = random_tris(1, 3)
    same = np.repeat(t, 9)
    nodes = assert_same_tree(same, 0, 9)
    assert len(nodes) == 1 and nodes["count"][0] == 9
    # a flat grid (all z equal) exercises empty-side NaN costs on one axis
    pos, idx, _, _ = procedural.grid(20, 20)
    assert_same_tree(host.make_tris(pos, idx), 0, 800)
    # ties between candidate planes: a symmetric mesh
    pos, idx, _, _ = procedural.box()
    nodes = assert_same_tree(host.make_tris(pos, idx), 0, 12)
    assert len(nodes) == 11


def test_tree_is_a_valid_bvh():
    tris = random_tris(3000, 11)
    nodes, order = host.bvh_build(tris, 0, 3000)
    assert sorted(order.tolist()) == list(range(3000))
    seen = np.zeros(3000, bool)
    stack = [0]
    while stack:
        n = nodes[stack.pop()]
        if n["count"] > 0:
            ids = order[n["first"]: n["first"] + n["count"]]
            assert not seen[ids].any()
            seen[ids] = True
            v = tris["v"][ids][:, :, :3].reshape(-1, 3)
            assert (v.min(0) >= n["aabb"][:3]).all() and (v.max(0) <= n["aabb"][3:]).all()
        else:
            for c in (n["first"], n["first"] + 1):
                assert c > 0 and (nodes["aabb"][c][:3] >= n["aabb"][:3]).all() and (nodes["aabb"][c][3:] <= n["aabb"][3:]).all()
                stack.append(c)
    assert seen.all()


def test_builder_bit_exact_on_reference_assets(reference_resources):
    for name in ("Box.gltf", "suzanne.glb", "simple_with_camera.glb", "simple_terrain.glb"):
        for pos, idx in gltf_mini.triangle_primitives(f"{reference_resources}/{name}"):
            tris = host.make_tris(pos, idx)
            assert_same_tree(tris, 0, len(tris))


def test_scene_assembly_matches_oracle_build_primitive():
    """add_primitive = Tri::new + build + flatten + append + rebase (scene.rs:516-577)."""
    s = host.Scene()
    meshes = [procedural.uv_sphere(12, 9), procedural.grid(9, 7, (3, 2)), procedural.box()]
    for k, (pos, idx, nrm, uv) in enumerate(meshes):
        s.add_primitive(pos, idx, normals=nrm, uv0=uv, transform=host.trs((k, 0, 0), rot_z=0.3 * k))
    a = s.arrays()
    tris = np.concatenate([host.make_tris(m[0], m[1]) for m in meshes])
    exts = np.zeros(len(tris), dtype=L.TRI_EXT)
    nodes_all, roots, off = [], [], 0
    for pos, idx, _, _ in meshes:
        n = len(idx) // 3
        root = sum(len(x) for x in nodes_all)
        nodes, order = orc.build_primitive(tris, exts, off, n, root)
        assert np.array_equal(a["tri_order"][off:off + n], order + off)
        nodes_all.append(nodes)
        roots.append(root)
        off += n
    assert a["nodes"].view(np.uint8).tobytes() == np.concatenate(nodes_all).tobytes()
    assert a["tris"].tobytes() == tris.tobytes()
    assert list(a["prims"]["bvh_idx"]) == roots


def test_tri_new_centroid_is_a_true_division():
    p = np.array([[0.1, 0.2, 0.3], [1.7, -2.9, 4.1], [1e-3, 5.5, -7.25]], np.float32)
    t = host.tri_new(*p)
    want = ((p[0] + p[1]) + p[2]) / np.float32(3.0)
    assert np.array_equal(t["v"][:, 3], want)
    assert np.array_equal(t["v"][:, :3], p)


def test_pack_oct_alpha_and_inverse_against_oracle():
    rng = np.random.default_rng(2)
    for n in rng.normal(size=(200, 3)):
        n = (n / np.linalg.norm(n)).astype(np.float32)
        assert np.array_equal(host.pack_oct(n), orc.pack_oct(n))
    assert np.array_equal(host.pack_oct([0, 0, 1]), [0.5, 0.5])
    for mode, cut in [(0, 0.5), (1, 0.5), (2, 0.25), (1, 7.0), (1, -1.0)]:
        assert host.pack_alpha(mode, cut) == int(orc.lib().orc_pack_alpha(mode, cut))
    assert host.pack_alpha(1, 0.5) == 1 | (32767 << 16)
    for _ in range(50):
        m = host.trs(rng.normal(size=3), rot_z=rng.normal(), scale=rng.uniform(0.2, 3, 3), rot_y=rng.normal(), rot_x=rng.normal())
        i64, i32 = host.mat4_inverse_f64(m), host.mat4_inverse_f32(m)
        np.testing.assert_allclose(i64, orc.mat4_inverse_f64(m), rtol=2e-7, atol=1e-7)
        np.testing.assert_allclose(i32, i64, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(i64.reshape(4, 4).T @ m.reshape(4, 4).T, np.eye(4), atol=1e-5)


def test_default_camera_and_uniform_block_layout():
    cam = host.Camera.default().to_gpu()
    want = orc.camera_default().view(L.GPU_CAMERA)[0]
    assert cam.tobytes() == want.tobytes()
    np.testing.assert_allclose(cam["dir"], [np.sqrt(0.5), 0, -np.sqrt(0.5)], atol=1e-7)
    assert tuple(cam["origin"]) == (-5.0, 0.0, 5.0) and cam["focus"] == 1.0

    s = procedural.box_scene(env_size=(8, 4))
    u = s.frame_uniforms((640, 480), frame=9, reject_hist=1, time=1.5)
    raw = np.frombuffer(u.tobytes(), np.uint8)
    assert raw.size == 608 and not raw[:512].any()                      # lights: zero, never read
    assert raw[512:560].tobytes() == cam.tobytes()                      # camera @512
    words = raw[560:608].view(np.uint32)
    assert list(words[:4]) == [12, 0, 0, 0]                             # tri_count @560
    assert list(words[4:7]) == [640, 480, 9]                            # res @576, frame @584
    assert raw[588:592].view(np.float32)[0] == 1.5                      # time @588
    assert list(words[8:12]) == [1, 11, 1, 0]                           # reject_hist, node_count, prim_count, pad


def test_instancing_textures_and_limits():
    s = host.Scene()
    pos, idx, nrm, uv = procedural.box()
    p0 = s.add_primitive(pos, idx, normals=nrm, uv0=uv, transform=host.trs((1, 2, 3)))
    p1 = s.add_instance(p0, host.trs((4, 5, 6), scale=(2, 1, 0.5)))
    a = s.arrays()
    assert p1 == 1 and len(a["tris"]) == 12 and a["prims"]["bvh_idx"][1] == a["prims"]["bvh_idx"][0]
    assert a["prims"]["material"][1].tobytes() == a["prims"]["material"][0].tobytes()
    np.testing.assert_allclose(a["prims"]["inv_transform"][1], host.mat4_inverse_f32(host.trs((4, 5, 6), scale=(2, 1, 0.5))))
    img = np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4)
    ref = s.add_texture(img)
    assert int(ref["offset"]) == 0 and int(ref["size"]) == (3 << 16) | 2
    tex = s.texture_data
    assert tex[0] == (0 << 24) | (1 << 16) | (2 << 8) | 3 and tex[5] == (20 << 24) | (21 << 16) | (22 << 8) | 23
    ref2 = s.add_texture(img)
    assert int(ref2["offset"]) == 6
    with pytest.raises(host.RtHostError):
        s.add_primitive(pos, idx + 100)           # vertex index out of range
    with pytest.raises(host.RtHostError):
        s.add_instance(7, host.IDENTITY)


def test_missing_normals_stay_zero_and_uv_sets_are_independent():
    s = host.Scene()
    pos, idx, nrm, uv = procedural.grid(2, 2)
    s.add_primitive(pos, idx, uv1=uv * 2)
    e = s.tri_exts
    assert not e["v"]["normal"].any() and not e["v"]["tc"][:, :, 0].any()
    assert e["v"]["tc"][:, :, 1].max() == 2.0


def test_focus_camera_casts_along_forward():
    s = procedural.box_scene(env_size=(8, 4))
    assert s.focus_camera(0)
    want = np.sqrt(50.0) - 0.5 * np.sqrt(2.0)      # camera (-5,0,5) looking at the cube edge ... along (1,0,-1)/sqrt2
    assert abs(s.camera(0).c.focus - want) < 0.3
    s2 = host.Scene()
    pos, idx, nrm, uv = procedural.box()
    s2.add_primitive(pos, idx, transform=host.trs((0, 50, 0)))
    s2.add_camera(host.Camera.default())
    assert not s2.focus_camera(0) and s2.camera(0).c.focus == 1.0


def test_scene_cache_round_trip(tmp_path):
    """.rtscene: every array and camera comes back byte for byte; damaged files are refused."""
    from raytracer_amd import host, procedural
    s = procedural.material_test_scene(tex=16, env_size=(32, 16))
    path = str(tmp_path / "m.rtscene")
    s.save(path)
    t = host.Scene.from_cache(path)
    a, b = s.arrays(), t.arrays()
    assert set(a) == set(b)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
    assert t.n_cameras == s.n_cameras
    u0 = np.ascontiguousarray(s.frame_uniforms((64, 48), frame=3)).tobytes()
    u1 = np.ascontiguousarray(t.frame_uniforms((64, 48), frame=3)).tobytes()
    assert u0 == u1
    raw = bytearray(open(path, "rb").read())
    # truncated
    open(path, "wb").write(raw[:-16])
    with pytest.raises(host.RtHostError):
        host.Scene.from_cache(path)
    # wrong magic
    bad = bytearray(raw)
    bad[0:8] = b"NOTSCENE"
    open(path, "wb").write(bad)
    with pytest.raises(host.RtHostError):
        host.Scene.from_cache(path)
    # a node that points outside the node array: first inner node's `first` field
    nodes = a["nodes"]
    inner = int(np.flatnonzero(nodes["count"] == 0)[0])
    off = 64 + ((48 * a["tris"].shape[0] + 15) & ~15) + ((96 * a["tris"].shape[0] + 15) & ~15) + 32 * inner + 24
    bad = bytearray(raw)
    bad[off:off + 4] = np.array([0x7FFFFFF0], np.uint32).tobytes()
    open(path, "wb").write(bad)
    with pytest.raises(host.RtHostError):
        host.Scene.from_cache(path)


def test_baked_reference_fixtures_load_and_match_the_mounted_assets(reference_resources):
    """tests/golden/ref_*.rtscene.gz are what this build's host pipeline makes of the reference's
    bundled scenes (tests/tools/bake_reference_assets.py); with the reference mounted, loading the
    assets again gives the same bytes, and trail.hdr decodes to the floats the RGBE fixture gives."""
    from util import REFERENCE_SCENES, load_reference_scene, reference_env
    names = {"box": "Box.gltf", "suzanne": "suzanne.glb", "simple_terrain": "simple_terrain.glb",
             "simple_with_camera": "simple_with_camera.glb"}
    assert set(names) == set(REFERENCE_SCENES)
    for key, fname in names.items():
        baked, a = load_reference_scene(key, env=False)
        live = host.Scene.from_path(os.path.join(reference_resources, fname))
        b = live.arrays()
        for k in ("tris", "tri_exts", "nodes", "prims", "texels", "tri_order"):
            assert a[k].tobytes() == b[k].tobytes(), (key, k)
        assert baked.n_cameras == live.n_cameras == 1
        assert bytes(baked.camera(0).c) == bytes(live.camera(0).c)
    s = host.Scene()
    s.set_equirectangular_env_map(os.path.join(reference_resources, "trail.hdr"))
    assert np.array_equal(s.env_map_data.view(np.uint32), reference_env().view(np.uint32))


def test_baked_reference_fixtures_are_self_consistent():
    """Without the reference: the fixtures load, have the documented shapes, and the oracle still
    reproduces the pinned primary hits of configuration 1 (Box.gltf, 512x512, frame 1)."""
    from oracle import orc
    from util import GOLDEN, load_reference_scene, oracle_scene
    shapes = {"box": (12, 1), "suzanne": (15744, 1), "simple_terrain": (8198, 2), "simple_with_camera": (970, 2)}
    for key, (n_tris, n_prims) in shapes.items():
        s, a = load_reference_scene(key, env=False)
        assert a["tris"].shape[0] == n_tris and a["prims"].shape[0] == n_prims and s.n_cameras == 1
    s, a = load_reference_scene("box")
    assert a["env"].shape == (512, 1024, 4)
    u = s.frame_uniforms((512, 512), frame=1, reject_hist=1)
    r = orc.render(oracle_scene(a), u, first_frame=1, n_frames=1, taps=True, n_threads=4)
    pinned = np.load(os.path.join(GOLDEN, "ref_box_config1_taps.npz"))
    assert np.array_equal(r["tri"], pinned["hit_tri"].astype(np.int32))
    assert np.array_equal(r["prim"], pinned["hit_prim"].astype(np.int32))
    assert r["counters"]["rays"] == int(pinned["rays"])
