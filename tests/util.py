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
