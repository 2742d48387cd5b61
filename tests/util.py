"""Shared helpers for the parity tests."""
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
