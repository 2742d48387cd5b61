"""glTF / PNG / Radiance loaders of the host pipeline (scene.rs:258-615 semantics)."""
import io
import os
import struct

import numpy as np
import pytest
from PIL import Image

from raytracer_amd import host, layouts as L, procedural
from glb_writer import Glb, png_bytes
import gltf_mini


def quad():
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 2.0], np.float32), (4, 1))       # not unit: the loader normalises
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    return pos, nrm, uv, idx


def build_test_glb():
    g = Glb()
    pos, nrm, uv, idx = quad()
    tex = g.image_texture(procedural.texture_checker(4, 2))
    m_tex = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": tex, "texCoord": 1},
                                             "baseColorFactor": [0.5, 0.6, 0.7, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.75},
                       alphaMode="MASK", alphaCutoff=0.3, emissiveFactor=[1.0, 0.5, 0.25],
                       extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}})
    a_pos = g.accessor(pos, "VEC3", 5126)
    a_nrm = g.accessor(nrm, "VEC3", 5126)
    a_uv0 = g.accessor(uv, "VEC2", 5126)
    a_uv1 = g.accessor((uv * 65535).astype(np.uint16), "VEC2", 5123, normalized=True)
    a_idx16 = g.accessor(idx, "SCALAR", 5123)
    a_idx32 = g.accessor(idx.astype(np.uint32)[::-1].copy(), "SCALAR", 5125)
    a_col = g.accessor(np.array([[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [9, 8, 7, 6]], np.uint8), "VEC4", 5121, normalized=True)
    mesh = g.mesh([
        {"attributes": {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv0, "TEXCOORD_1": a_uv1, "COLOR_0": a_col},
         "indices": a_idx16, "material": m_tex},
        {"attributes": {"POSITION": a_pos}, "indices": a_idx32},          # no material, no normals
    ])
    child = g.node(root=False, mesh=mesh, translation=[0, 0, 2])           # instance of the same mesh
    g.node(mesh=mesh, translation=[1, 2, 3], rotation=[0, 0.70710678, 0, 0.70710678], scale=[2, 2, 2], children=[child])
    g.doc["cameras"].append({"type": "perspective", "perspective": {"yfov": 0.5, "aspectRatio": 1.5, "znear": 0.1}})
    g.node(camera=0, translation=[0, 1, 5])
    return g.tobytes()


def test_glb_scene_graph_materials_and_instancing():
    s = host.Scene()
    s.add_gltf_bytes(build_test_glb())
    a = s.arrays()
    pr = a["prims"]
    assert len(pr) == 4 and len(a["tris"]) == 4                     # 2 primitives + 2 instanced copies, 2 triangles each
    assert list(pr["bvh_idx"]) == [0, 1, 0, 1]
    m = pr["material"][0]
    np.testing.assert_allclose(m["albedo_factor"], [0.5, 0.6, 0.7, 1.0])
    assert m["metallic_factor"] == np.float32(0.25) and m["roughness_factor"] == np.float32(0.75)
    np.testing.assert_allclose(m["emissive_factor"], [4.0, 2.0, 1.0])
    assert int(m["alpha_settings"]) == 1 | (int(np.float32(0.3) * np.float32(65535.0)) << 16)
    assert int(m["albedo"]["size"]) == (4 << 16) | 4 and int(m["albedo_texcoord"]) == 1
    d = pr["material"][1]                                           # gltf-crate defaults
    assert d["metallic_factor"] == 1.0 and d["roughness_factor"] == 1.0 and int(d["alpha_settings"]) == 32767 << 16
    assert tuple(d["albedo_factor"]) == (1, 1, 1, 1) and int(d["albedo"]["size"]) == 0
    # texel packing R<<24 | G<<16 | B<<8 | A, rows top first
    chk = procedural.texture_checker(4, 2)
    want = (chk[..., 0].astype(np.uint32) << 24) | (chk[..., 1].astype(np.uint32) << 16) | (chk[..., 2].astype(np.uint32) << 8) | chk[..., 3]
    assert np.array_equal(a["texels"], want.reshape(-1))
    # vertex data of primitive 0: only UV set 1 is referenced by a texture, so set 0 stays zero
    e = a["tri_exts"][:2]
    assert not e["v"]["tc"][:, :, 0].any()
    assert set(np.unique(e["v"]["tc"][:, :, 1])) == {0.0, 1.0}
    assert np.allclose(e["v"]["normal"], host.pack_oct([2.0, 0, 0] / np.float32(2.0)))   # gltf +Z -> world +X
    assert set(e["v"]["color"].reshape(-1)) <= {0xFF0000FF, 0x00FF00FF, 0x0000FFFF, 0x09080706}
    assert not a["tri_exts"][2:]["v"]["normal"].any()
    # positions are re-based (x, y, z) -> (z, x, y) and stay primitive-local
    v = a["tris"]["v"][:2, :, :3].reshape(-1, 3)
    assert v[:, 0].max() == 0.0 and v[:, 1].max() == 1.0 and v[:, 2].max() == 1.0
    # node transform: T * R * S, then C * M * C^T; child transforms compose
    t0 = pr["transform"][0].reshape(4, 4).T
    np.testing.assert_allclose(t0[:3, 3], [3, 1, 2], atol=1e-6)
    np.testing.assert_allclose(np.linalg.det(t0[:3, :3]), 8.0, rtol=1e-5)
    t2 = pr["transform"][2].reshape(4, 4).T
    gl = np.array([[0, 0, 2, 1], [0, 2, 0, 2], [-2, 0, 0, 3], [0, 0, 0, 1.0]]) @ np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 2], [0, 0, 0, 1.0]])
    C = np.array([[0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    np.testing.assert_allclose(t2, C @ gl @ C.T, atol=1e-5)
    np.testing.assert_allclose(pr["inv_transform"][0].reshape(4, 4).T @ t0, np.eye(4), atol=1e-5)
    np.testing.assert_allclose(pr["inv_transform"][2], host.mat4_inverse_f32(pr["transform"][2]), rtol=1e-6, atol=1e-7)
    # camera: position re-based, yfov kept, focus 50 (input.rs:242-250)
    assert s.n_cameras == 1
    c = s.camera(0).c
    np.testing.assert_allclose(list(c.position), [5, 0, 1], atol=1e-6)
    assert abs(c.fovy - 0.5) < 1e-7 and c.focus == 50.0 and abs(c.aspect - 1.5) < 1e-7


def test_gltf_json_with_data_uri_and_errors(tmp_path):
    import base64
    import json
    pos, nrm, uv, idx = quad()
    blob = pos.tobytes() + idx.tobytes()
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "matrix": [1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 1]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 12}],
           "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]}
    p = tmp_path / "quad.gltf"
    p.write_text(json.dumps(doc))
    s = host.Scene.from_path(str(p))
    assert len(s.tris) == 2 and s.n_cameras == 1                       # default camera added (scene.rs:698-702)
    doc["meshes"][0]["primitives"][0]["mode"] = 1                       # LINES: the reference panics, we report
    with pytest.raises(host.RtHostError):
        host.Scene().add_gltf_bytes(json.dumps(doc).encode())
    del doc["meshes"][0]["primitives"][0]["mode"]
    del doc["meshes"][0]["primitives"][0]["indices"]                    # non-indexed: the reference panics, we report
    with pytest.raises(host.RtHostError):
        host.Scene().add_gltf_bytes(json.dumps(doc).encode())
    with pytest.raises(host.RtHostError):
        host.Scene().add_gltf_bytes(b"{ not json")
    with pytest.raises(host.RtHostError):
        host.Scene.from_path(str(tmp_path / "missing.glb"))


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P", "I;16", "1"])
def test_png_decoder_matches_pil(mode):
    rng = np.random.default_rng(3)
    base = Image.fromarray(rng.integers(0, 256, (13, 17, 4), dtype=np.uint8), "RGBA")
    if mode == "I;16":
        img = Image.fromarray(rng.integers(0, 65536, (13, 17), dtype=np.uint16))
    elif mode == "P":
        img = base.convert("RGB").quantize(16)
    else:
        img = base.convert(mode)
    buf = io.BytesIO()
    img.save(buf, "PNG")
    g = Glb()
    pos, nrm, uv, idx = quad()
    g.doc["images"].append({"bufferView": g._view(buf.getvalue()), "mimeType": "image/png"})
    g.doc["textures"].append({"source": 0})
    mat = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": 0}})
    mesh = g.mesh([{"attributes": {"POSITION": g.accessor(pos, "VEC3", 5126), "TEXCOORD_0": g.accessor(uv, "VEC2", 5126)},
                    "indices": g.accessor(idx, "SCALAR", 5123), "material": mat}])
    g.node(mesh=mesh)
    s = host.Scene()
    s.add_gltf_bytes(g.tobytes())
    t = s.texture_data
    got = np.stack([(t >> 24) & 255, (t >> 16) & 255, (t >> 8) & 255, t & 255], -1).reshape(13, 17, 4)
    if mode == "I;16":
        v16 = np.asarray(img).astype(np.uint32)
        want = np.stack([(v16 * 255 + 32767) // 65535] * 3 + [np.full_like(v16, 255)], -1)
    else:
        want = np.asarray(img.convert("RGBA"))
    assert np.array_equal(got, want)


def _texture_through_gltf(image_bytes, mime, w, h):
    g = Glb()
    pos, nrm, uv, idx = quad()
    g.doc["images"].append({"bufferView": g._view(image_bytes), "mimeType": mime})
    g.doc["textures"].append({"source": 0})
    mat = g.material(pbrMetallicRoughness={"baseColorTexture": {"index": 0}})
    g.node(mesh=g.mesh([{"attributes": {"POSITION": g.accessor(pos, "VEC3", 5126), "TEXCOORD_0": g.accessor(uv, "VEC2", 5126)},
                         "indices": g.accessor(idx, "SCALAR", 5123), "material": mat}]))
    s = host.Scene()
    s.add_gltf_bytes(g.tobytes())
    t = s.texture_data
    return np.stack([(t >> 24) & 255, (t >> 16) & 255, (t >> 8) & 255, t & 255], -1).reshape(h, w, 4)


def _photo(w, h, seed=5):
    """Smooth colour gradients plus noise and a hard edge: exercises every DCT coefficient."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 110 * np.cos((x + y) / 9.0),
                    255.0 * ((x // 6 + y // 4) % 2)], -1) + rng.normal(0, 12, (h, w, 3))
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")


@pytest.mark.parametrize("kw", [
    dict(subsampling=0), dict(subsampling=1), dict(subsampling=2),
    dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True, quality=95),
    dict(subsampling=2, quality=30, optimize=True), dict(gray=True), dict(gray=True, progressive=True),
    dict(subsampling=1, restart_marker_blocks=3), dict(subsampling=2, size=(1, 1)), dict(subsampling=2, size=(3, 2)),
    dict(subsampling=2, size=(16, 8)), dict(subsampling=1, size=(250, 3)), dict(subsampling="4:1:1"),
])
def test_jpeg_decoder_matches_libjpeg(kw):
    """Baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, grayscale, restart intervals, odd sizes: the
    texels equal those of the libjpeg family (PIL), whose arithmetic the decoder follows."""
    kw = dict(kw)
    w, h = kw.pop("size", (61, 45))
    img = _photo(w, h)
    if kw.pop("gray", False):
        img = img.convert("L")
    buf = io.BytesIO()
    img.save(buf, "JPEG", **{"quality": 85, **kw})
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA")).astype(np.int32)
    got = _texture_through_gltf(buf.getvalue(), "image/jpeg", w, h).astype(np.int32)
    assert got.shape == want.shape and np.all(got[..., 3] == 255)
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_jpeg_errors_are_reported():
    buf = io.BytesIO()
    _photo(32, 32).save(buf, "JPEG")
    data = buf.getvalue()
    with pytest.raises(host.RtHostError, match="JPEG"):
        _texture_through_gltf(data[:2] + b"\xff\xc9\x00\x04\x00\x00" + data[2:], "image/jpeg", 32, 32)  # arithmetic coding
    with pytest.raises(host.RtHostError, match="JPEG"):
        _texture_through_gltf(data[:20], "image/jpeg", 32, 32)
    cmyk = io.BytesIO()
    _photo(16, 16).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(host.RtHostError, match="three-component"):
        _texture_through_gltf(cmyk.getvalue(), "image/jpeg", 16, 16)


def rgbe_encode(rgb):
    m = rgb.max(-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0)
    scale = np.where(m > 1e-32, 256.0 / np.exp2(e), 0)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255)
    out[..., 3] = np.where(m > 1e-32, e + 128, 0)
    return out


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr_decoder(tmp_path, rle):
    rng = np.random.default_rng(5)
    w, h = 16, 6
    rgbe = rgbe_encode(rng.uniform(0, 40, (h, w, 3)) * (rng.random((h, w, 1)) > 0.1))
    body = b""
    for y in range(h):
        if rle:
            body += bytes([2, 2, w >> 8, w & 255])
            for c in range(4):
                row = rgbe[y, :, c]
                body += bytes([128 + 4, row[0]]) if (row[:4] == row[0]).all() else bytes([4]) + row[:4].tobytes()
                body += bytes([w - 4]) + row[4:].tobytes()
        else:
            body += rgbe[y].tobytes()
    p = tmp_path / "env.hdr"
    p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + body)
    s = host.Scene()
    s.set_equirectangular_env_map(str(p))
    env = s.env_map_data
    want = rgbe[..., :3].astype(np.float32) * np.exp2(rgbe[..., 3:].astype(np.float32) - 136.0)
    want[rgbe[..., 3] == 0] = 0
    assert env.shape == (h, w, 4) and np.array_equal(env[..., :3], want) and (env[..., 3] == 1).all()
    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"P6\n")
    with pytest.raises(host.RtHostError):
        s.set_equirectangular_env_map(str(bad))


def test_reference_assets_load_with_expected_shapes(reference_resources):
    """Triangle / primitive counts parsed independently from the files' JSON chunks."""
    expect = {"Box.gltf": (12, 1, 0), "suzanne.glb": (15744, 1, 0), "simple_with_camera.glb": (970, 2, 1),
              "simple_terrain.glb": (8198, 2, 1)}
    for name, (tris, prims, cams) in expect.items():
        s = host.Scene()
        raw = open(f"{reference_resources}/{name}", "rb").read()
        s.add_gltf_bytes(raw, base_dir=reference_resources)
        a = s.arrays()
        assert (len(a["tris"]), len(a["prims"]), s.n_cameras) == (tris, prims, cams)
        # positions agree with an independent reader, up to the BVH reordering
        ref = np.concatenate([host.make_tris(p, i) for p, i in gltf_mini.triangle_primitives(f"{reference_resources}/{name}")])
        assert sorted(map(bytes, a["tris"])) == sorted(map(bytes, ref))
    s = host.Scene.from_path(f"{reference_resources}/Box.gltf", f"{reference_resources}/trail.hdr")
    env = s.env_map_data
    assert env.shape == (512, 1024, 4) and env[..., :3].min() >= 0 and np.isfinite(env).all()
    m = s.primitives["material"][0]
    np.testing.assert_allclose(m["albedo_factor"], [0.8, 0, 0, 1])
    assert m["metallic_factor"] == 0.0


def test_hostile_sizes_are_refused_not_followed(tmp_path):
    """Counts and offsets a file supplies are checked by subtraction before they are used: a
    negative or wrapping value is a format error, never an out-of-bounds read, an exception across
    the C ABI or an attempt to allocate what the file claims."""
    import json
    from glb_writer import Glb

    def tampered(edit):
        g = Glb()
        pos, nrm, uv, idx = quad()
        a_pos = g.accessor(pos, "VEC3", 5126)
        a_idx = g.accessor(idx, "SCALAR", 5123)
        g.node(mesh=g.mesh([{"attributes": {"POSITION": a_pos}, "indices": a_idx}]))
        edit(g.doc)
        return g.tobytes()

    for edit in (lambda d: d["accessors"][0].update(byteOffset=-16),
                 lambda d: d["accessors"][0].update(count=2 ** 62),
                 lambda d: d["bufferViews"][0].update(byteOffset=2 ** 63 - 8),
                 lambda d: d["bufferViews"][0].update(byteStride=2 ** 61),
                 lambda d: d["accessors"][1].update(count=-3)):
        s = host.Scene()
        with pytest.raises(host.RtHostError):
            s.add_gltf_bytes(tampered(edit))

    # .rtscene header that claims 2^60 triangles
    p = tmp_path / "bad.rtscene"
    p.write_bytes(b"RTSCENE1" + struct.pack("<3Q4I", 2 ** 60, 1, 0, 1, 0, 0, 0) + bytes(16) + bytes(64))
    with pytest.raises(host.RtHostError, match="beyond what a scene can hold|wrong length"):
        host.Scene.from_cache(str(p))

    # Radiance file whose resolution line promises far more than the file holds
    q = tmp_path / "bad.hdr"
    q.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 60000 +X 60000\n" + bytes(64))
    s = host.Scene()
    with pytest.raises(host.RtHostError, match="does not fit"):
        s.set_equirectangular_env_map(str(q))
