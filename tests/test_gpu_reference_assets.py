"""The reference's own bundled scenes (resources/*.gltf|glb + trail.hdr, baked into tests/golden by
tests/tools/bake_reference_assets.py) through the HIP path, against the CPU oracle on the same arrays.

Configurations 1 and 2 of the benchmark list are run exactly as named: Box.gltf 512x512 1 spp and
suzanne.glb 512x512 64 spp with trail.hdr, default camera (input.rs:202-219), reference
parameters (8 bounces, 50 % termination, aperture 0.025).  simple_terrain.glb and
simple_with_camera.glb bring a glTF camera (input.rs:222-253), TRS node transforms and two
primitives.  The last test renders a textured GLB that goes through the glTF + PNG loaders.
"""
import os

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import device, host, procedural
from util import GOLDEN, load_reference_scene, oracle_params, oracle_scene, psnr, tonemapped

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "staged"])
def pipeline(request, monkeypatch):
    """Every test of this module runs twice: as the library chooses (small frame groups = one
    persistent launch, k_path) and with every group forced through the staged pipeline."""
    if request.param == "staged":
        monkeypatch.setenv("RTHIP_FORCE_STAGED", "1")
    return request.param

THREADS = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def ctx():
    c = device.Context(0)
    yield c
    c.close()


def both(ctx, scene, arrays, res, first=1, n=1, params=None, taps=True, camera_idx=0):
    p = params or device.default_params()
    p.flags |= device.FLAG_COUNTERS | (device.FLAG_TAPS if taps else 0)
    u = scene.frame_uniforms(res, frame=first, reject_hist=1, camera_idx=camera_idx)
    ctx.upload_scene(arrays)
    ctx.set_frame_uniforms(u)
    ctx.reset_counters()
    ctx.render(first, n, 1, p)
    got = dict(accum=ctx.read_accum(), counters=ctx.counters())
    if taps:
        got.update(ctx.read_taps())
    ref = orc.render(oracle_scene(arrays), u, oracle_params(p), first, n, 1, taps=taps, n_threads=THREADS)
    return got, ref


def check_frame(got, ref, hit_bar=0.9999, seed_bar=0.999):
    assert float((got["tri"] == ref["tri"]).mean()) >= hit_bar
    same_hit = got["tri"] == ref["tri"]
    assert np.array_equal(got["prim"][same_hit], ref["prim"][same_hit])
    np.testing.assert_allclose(got["t"][same_hit], ref["t"][same_hit], rtol=1e-5, atol=1e-6)
    assert float((got["seed"] == ref["seed"]).mean()) >= seed_bar
    # Same final RNG state = same branches taken.  The radiance may still differ on a handful of
    # pixels: trail.hdr is point-sampled (shader.wgsl:691-694) and has hard edges (the sun), and
    # atan2 / asin of the device library differ from the C library's in the last ulp, which moves a
    # lookup that falls on a texel boundary into the neighbouring texel.
    same = got["seed"] == ref["seed"]
    bad = ~np.isclose(got["accum"], ref["accum"], rtol=2e-3, atol=2e-4).all(axis=-1) & same
    assert bad.mean() <= 1e-4, int(bad.sum())
    assert np.array_equal(got["accum"][..., 3], ref["accum"][..., 3])
    for k in ("rays", "misses", "kills", "shade_calls"):
        assert abs(got["counters"][k] - ref["counters"][k]) <= max(4, 2e-3 * ref["counters"][k]), k


def test_config1_box_gltf_512_one_sample(ctx):
    s, a = load_reference_scene("box")
    assert a["tris"].shape[0] == 12 and a["prims"].shape[0] == 1
    got, ref = both(ctx, s, a, (512, 512))
    check_frame(got, ref)
    pinned = np.load(os.path.join(GOLDEN, "ref_box_config1_taps.npz"))
    assert np.array_equal(ref["tri"], pinned["hit_tri"].astype(np.int32))        # the oracle has not drifted
    assert float((got["tri"] == pinned["hit_tri"]).mean()) >= 0.9999
    assert got["counters"]["paths"] == 512 * 512
    # traversal work equals the oracle's, step for step
    for k in ("inner_steps", "tri_tests", "prims_visited"):
        assert abs(got["counters"][k] - ref["counters"][{"prims_visited": "root_tests"}.get(k, k)]) <= 8, k


def test_config2_suzanne_glb_512_64_samples(ctx):
    s, a = load_reference_scene("suzanne")
    assert a["tris"].shape[0] == 15744 and a["prims"].shape[0] == 1
    m = a["prims"]["material"][0]
    assert m["metallic_factor"] == 1.0 and m["roughness_factor"] == 1.0       # gltf-crate defaults, scene.rs:458-468
    got, ref = both(ctx, s, a, (512, 512))
    check_frame(got, ref)
    got, ref = both(ctx, s, a, (512, 512), n=64, taps=False)
    assert np.all(got["accum"][..., 3] == 64.0)
    p = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert p >= 45.0, p
    assert abs(got["counters"]["rays"] - ref["counters"]["rays"]) <= 2e-3 * ref["counters"]["rays"]
    # the default camera sees the head as a few hundred pixels; a second camera one head-width
    # away from it fills a third of the frame
    v = a["tris"]["v"][:, :, :3].reshape(-1, 3)
    c, e = (v.min(0) + v.max(0)) / 2, (v.max(0) - v.min(0)).max()
    s.add_camera(host.Camera.look((c[0] + e, c[1], c[2]), yaw=np.pi, pitch=0.0, fovy=np.radians(60.0), focus=e))
    got, ref = both(ctx, s, a, (512, 512), camera_idx=1)
    assert (ref["tri"] >= 0).mean() > 0.3
    check_frame(got, ref)
    got, ref = both(ctx, s, a, (512, 512), n=64, taps=False, camera_idx=1)
    p = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert p >= 45.0, p


def test_config2_as_worded_suzanne_with_sun_and_trail_hdr(ctx):
    """Configuration 2 as the benchmark definition words it: "sun + trail.hdr HDRI".  The live shader has no sun
    (the block is commented out, shader.wgsl:916-937), so this is the opt-in enable_sun = 1 against
    the oracle's restatement of that block, from a camera that fills the frame with the head."""
    s, a = load_reference_scene("suzanne")
    v = a["tris"]["v"][:, :, :3].reshape(-1, 3)
    c, e = (v.min(0) + v.max(0)) / 2, (v.max(0) - v.min(0)).max()
    s.add_camera(host.Camera.look((c[0] + e, c[1], c[2]), yaw=np.pi, pitch=0.0, fovy=np.radians(60.0), focus=e))
    p = device.default_params(enable_sun=1)
    got, ref = both(ctx, s, a, (512, 512), params=p, camera_idx=1)
    assert ref["counters"]["shadow_rays"] > 10000
    check_frame(got, ref)
    assert abs(got["counters"]["shadow_rays"] - ref["counters"]["shadow_rays"]) <= max(4, 2e-3 * ref["counters"]["shadow_rays"])
    p = device.default_params(enable_sun=1)
    got, ref = both(ctx, s, a, (512, 512), n=64, params=p, taps=False, camera_idx=1)
    assert np.all(got["accum"][..., 3] == 64.0)
    db = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert db >= 45.0, db
    # and the sun does light the head: brighter than the HDRI-only render of the same frames
    dark, _ = both(ctx, s, a, (512, 512), n=64, taps=False, camera_idx=1)
    assert float(got["accum"][..., :3].sum()) > 1.02 * float(dark["accum"][..., :3].sum())


@pytest.mark.parametrize("name,res", [("simple_terrain", (768, 384)), ("simple_with_camera", (640, 360))])
def test_scenes_with_gltf_camera_and_node_transforms(ctx, name, res):
    s, a = load_reference_scene(name)
    assert a["prims"].shape[0] == 2 and s.n_cameras == 1
    cam = s.camera(0).c
    assert cam.focus == 50.0 and cam.aspect > 1.0                              # input.rs:242-250
    got, ref = both(ctx, s, a, res)
    assert (ref["tri"] >= 0).mean() > 0.2                                      # the camera looks at the scene
    check_frame(got, ref)
    assert len(np.unique(got["prim"][got["tri"] >= 0])) == 2
    got, ref = both(ctx, s, a, (res[0] // 2, res[1] // 2), n=32, taps=False)
    p = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert p >= 45.0, p


def test_autofocus_ray_on_the_device_equals_the_host_ray_cast(ctx):
    """Scene::focus_camera (scene.rs:670-677) is one closest-hit ray along the camera's forward
    axis; the device answers it through rthip_trace_rays with the distance the host finds."""
    s, a = load_reference_scene("simple_terrain", env=False)
    ctx.upload_scene(a)
    g = s.camera(0).to_gpu()
    q = ctx.trace_rays(np.array(g["origin"], np.float32), np.array(g["dir"], np.float32))
    assert q["tri"][0] >= 0
    assert s.focus_camera(0)
    np.testing.assert_allclose(s.camera(0).c.focus, q["t"][0], rtol=1e-5)


def textured_glb():
    from glb_writer import Glb
    g = Glb()
    stone = g.image_texture(procedural.texture_stone(64, 3))
    leaves = g.image_texture(procedural.texture_leaves(64, 5))
    mr = g.image_texture(procedural.texture_metal_rough(32, 9))
    m_floor = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": stone}, "metallicFactor": 0.0, "roughnessFactor": 0.9})
    m_leaf = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": leaves, "texCoord": 1}, "metallicFactor": 0.0},
                        alphaMode="MASK", alphaCutoff=0.5, doubleSided=True)
    m_ball = g.material(pbrMetallicRoughness={"baseColorFactor": [0.9, 0.6, 0.3, 1.0], "metallicRoughnessTexture": {"index": mr}},
                        emissiveFactor=[0.2, 0.1, 0.0])
    m_glass = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": leaves}, "baseColorFactor": [1, 1, 1, 0.6]}, alphaMode="BLEND")

    def prim(mesh, material, uv1=False):
        pos, idx, nrm, uv = mesh
        att = {"POSITION": g.accessor(pos, "VEC3", 5126), "NORMAL": g.accessor(nrm, "VEC3", 5126),
               "TEXCOORD_0": g.accessor(uv, "VEC2", 5126)}
        if uv1:
            att["TEXCOORD_1"] = g.accessor((np.clip(uv, 0, 1) * 65535).astype(np.uint16), "VEC2", 5123, normalized=True)
        return {"attributes": att, "indices": g.accessor(idx.astype(np.uint32), "SCALAR", 5125), "material": material}

    # glTF is Y-up: the loader's axis change maps (x, y, z) -> (z, x, y)
    floor = g.mesh([prim(procedural.grid(12, 12, size=(8.0, 8.0), uv_scale=3.0), m_floor)])
    ball = g.mesh([prim(procedural.uv_sphere(24, 16, radius=0.8), m_ball)])
    fan = g.mesh([prim(procedural.grid(4, 4, size=(2.0, 2.0)), m_leaf, uv1=True), prim(procedural.grid(2, 2, size=(1.5, 1.5)), m_glass)])
    g.node(mesh=floor, rotation=[-0.70710678, 0, 0, 0.70710678])             # the XY grid laid flat (normal +Y)
    g.node(mesh=ball, translation=[0.0, 0.8, 0.0], scale=[1.0, 1.2, 1.0])
    g.node(mesh=ball, translation=[2.0, 0.5, -1.0], scale=[0.6, 0.6, 0.6])    # instance
    g.node(mesh=fan, translation=[-1.5, 1.0, 0.5], rotation=[0, 0.38268343, 0, 0.92387953])
    g.doc["cameras"].append({"type": "perspective", "perspective": {"yfov": 0.7, "aspectRatio": 1.5, "znear": 0.1}})
    g.node(camera=0, translation=[0.5, 1.6, 6.0], rotation=[-0.08715574, 0, 0, 0.9961947])
    return g.tobytes()


def test_textured_glb_through_the_loaders_and_the_device(ctx, tmp_path):
    """glTF + PNG loaders -> upload -> render: textures on both UV sets, MASK and BLEND alpha in
    the traversal, a metallic-roughness map, instancing and a glTF camera, from a file."""
    path = tmp_path / "scene.glb"
    path.write_bytes(textured_glb())
    s = host.Scene.from_path(str(path))
    s.set_env(procedural.sky_env(128, 64))
    s.focus_camera(0)
    a = s.arrays()
    assert a["prims"].shape[0] == 5 and a["texels"].size == 64 * 64 * 2 + 32 * 32
    modes = sorted(int(x) & 3 for x in a["prims"]["material"]["alpha_settings"])
    assert modes == [0, 0, 0, 1, 2]
    got, ref = both(ctx, s, a, (384, 256), n=2)
    assert (ref["tri"] >= 0).mean() > 0.5
    check_frame(got, ref, seed_bar=0.998)
    assert ref["counters"]["alpha_tests"] > 10000
    assert abs(got["counters"]["alpha_tests"] - ref["counters"]["alpha_tests"]) <= 2e-3 * ref["counters"]["alpha_tests"]
    got, ref = both(ctx, s, a, (192, 128), n=32, taps=False)
    p = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    assert p >= 45.0, p
