"""Shared helpers for the parity tests."""
import os

import numpy as np

from oracle import orc


def oracle_scene(arrays):
    env = arrays["env"]
    return orc.make_scene(arrays["tris"], arrays["tri_exts"], arrays["nodes"], arrays["prims"],
                          arrays["texels"], env, env.shape[1], env.shape[0])


def oracle_params(p):
    """rthip RenderParams -> oracle Params (same fields, same meaning)."""
    return orc.Params(p.max_bounces, p.term_prob, p.aperture_radius, p.stack_depth, p.enable_sun)


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def tonemapped(accum):
    """Display-referred image in [0,1] (PBR Neutral + sRGB), the space PSNR is quoted in."""
    return orc.resolve(accum).astype(np.float64)[..., :3] / 255.0


# ---- the reference's own assets, baked by tests/tools/bake_reference_assets.py ---------------------

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFERENCE_SCENES = {"box": "ref_box.rtscene.gz", "suzanne": "ref_suzanne.rtscene.gz",
                    "simple_terrain": "ref_simple_terrain.rtscene.gz",
                    "simple_with_camera": "ref_simple_with_camera.rtscene.gz"}


def reference_env():
    """resources/trail.hdr as the rgba32f rows the loader produces: mantissa * 2^(e - 136)."""
    rgbe = np.load(os.path.join(GOLDEN, "ref_trail_rgbe.npz"))["rgbe"]
    e = rgbe[..., 3].astype(np.int32)
    out = np.ones(rgbe.shape[:2] + (4,), np.float32)
    out[..., :3] = np.ldexp(rgbe[..., :3].astype(np.float32), (e - 136)[..., None])
    out[..., :3][e == 0] = 0.0
    return out


def load_reference_scene(name, env=True):
    """(Scene, arrays) of one of the reference's bundled scenes, read through the .rtscene loader
    (Scene.from_cache); env=True adds trail.hdr as scene.rs:603-615 would."""
    import gzip
    import tempfile
    from raytracer_amd import host
    blob = gzip.open(os.path.join(GOLDEN, REFERENCE_SCENES[name]), "rb").read()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, name + ".rtscene")
        with open(path, "wb") as f:
            f.write(blob)
        scene = host.Scene.from_cache(path)
    if env:
        scene.set_env(reference_env())
    else:
        scene.set_env(np.ones((2, 4, 4), np.float32))
    return scene, scene.arrays()


def tiled_reference_terrain(tiles, top_view=False):
    """Configuration 5: the reference's simple_terrain.glb replicated on a tiles x tiles grid, the
    copies pre-transformed to world space and merged into ONE primitive with ONE BVH (78 x 78 gives
    49.9 M triangles); tile pitch = the scene's bounding-box extent.  Returns (Scene, arrays).  The
    first camera is the file's own glTF camera; top_view adds a second one that looks down on the
    whole tiling."""
    from raytracer_amd import host
    base, a = load_reference_scene("simple_terrain", env=False)
    # world-space vertices of the two primitives' triangles (column-major transform, row vectors)
    world = []
    for p in a["prims"]:
        root = int(p["bvh_idx"])
        # the triangles of a primitive are the leaves under its root: collect them by walking the tree
        stack, tri_ids = [root], []
        nodes = a["nodes"]
        while stack:
            n = nodes[stack.pop()]
            if n["count"] > 0:
                tri_ids.extend(range(int(n["first"]), int(n["first"]) + int(n["count"])))
            else:
                stack += [int(n["first"]), int(n["first"]) + 1]
        v = a["tris"]["v"][np.array(sorted(tri_ids)), :, :3].astype(np.float64)
        m = p["transform"].reshape(4, 4).T.astype(np.float64)
        world.append(v @ m[:3, :3].T + m[:3, 3])
    tri = np.concatenate(world)                                   # (8198, 3, 3)
    lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    pitch = hi - lo
    ix, iy = np.meshgrid(np.arange(tiles), np.arange(tiles), indexing="ij")
    offs = np.stack([(ix.ravel() - (tiles - 1) / 2) * pitch[0], (iy.ravel() - (tiles - 1) / 2) * pitch[1], np.zeros(tiles * tiles)], -1)
    pos = (tri[None].astype(np.float32) + offs[:, None, None, :].astype(np.float32)).reshape(-1, 3)
    idx = np.arange(pos.shape[0], dtype=np.uint32)
    # flat shading normals from the geometry
    e1, e2 = pos[1::3] - pos[0::3], pos[2::3] - pos[0::3]
    n = np.cross(e1, e2)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    nrm = np.repeat(n.astype(np.float32), 3, axis=0)
    s = host.Scene()
    s.add_primitive(pos, idx, normals=nrm, material=np.array(a["prims"]["material"][0]))
    s.add_camera(base.camera(0))
    if top_view:
        fovy = np.radians(60.0)
        height = (tiles * pitch[0] / 2) / (np.tan(fovy / 2) * 16.0 / 9.0)
        s.add_camera(host.Camera.look((-height * np.tan(np.radians(5.0)), 0.0, hi[2] + height), yaw=0.0, pitch=np.radians(-85.0), fovy=fovy, focus=height))
    s.set_env(np.ones((2, 4, 4), np.float32) * np.float32(0.7))
    return s, s.arrays()
