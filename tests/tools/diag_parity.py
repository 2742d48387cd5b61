"""Diagnostic: GPU vs oracle agreement per scene and bounce limit (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc
from raytracer_amd import device, procedural
from util import oracle_params, oracle_scene

ctx = device.Context(0)
for name, fn, kw, res in [("materials", procedural.material_test_scene, dict(tex=64, env_size=(128, 64)), (160, 96)),
                          ("blob", procedural.blob_scene, dict(env_size=(256, 128)), (128, 128))]:
    s = fn(**kw); a = s.arrays()
    ctx.upload_scene(a)
    for mb in (1, 2, 3, 8):
        p = device.default_params(max_bounces=mb, flags=device.FLAG_TAPS | device.FLAG_COUNTERS)
        u = s.frame_uniforms(res, frame=1, reject_hist=1)
        ctx.set_frame_uniforms(u); ctx.reset_counters(); ctx.render(1, 1, 1, p)
        got = dict(accum=ctx.read_accum(), counters=ctx.counters(), **ctx.read_taps())
        ref = orc.render(oracle_scene(a), u, oracle_params(p), 1, 1, 1, taps=True, n_threads=8)
        bad = got["seed"] != ref["seed"]
        print(name, "max_bounces", mb, "hit", (got["tri"] == ref["tri"]).mean(), "seed", 1 - bad.mean(),
              "prims of bad pixels", np.bincount(ref["prim"][bad] + 1, minlength=9)[:9],
              {k: (got["counters"][k], ref["counters"].get(k)) for k in ("rays", "inner_steps", "tri_tests", "alpha_tests")})
ctx.close()
