import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc
from raytracer_amd import device, procedural, host
from util import oracle_params, oracle_scene
ctx = device.Context(0)
s = procedural.material_test_scene(tex=64, env_size=(128, 64)); a = s.arrays()
ctx.upload_scene(a)
res = (128, 128)
for pos, yaw, pitch in [((1.0, 0.5, 0.001), 0.3, 0.6), ((-3.0, -2.0, 0.001), 0.0, 0.3), ((2.5, -0.5, 1.001), 2.0, 0.5), ((3.05, -0.3, 0.5), 0.4, -0.3)]:
    cam = host.Camera.look(pos, yaw=yaw, pitch=pitch, fovy=np.radians(120), focus=1.0)
    p = device.default_params(max_bounces=1, aperture_radius=0.0, flags=device.FLAG_TAPS | device.FLAG_COUNTERS)
    u = s.frame_uniforms(res, frame=1, reject_hist=1, gpu_camera=cam.to_gpu())
    ctx.set_frame_uniforms(u); ctx.reset_counters(); ctx.render(1, 1, 1, p)
    got = dict(counters=ctx.counters(), **ctx.read_taps())
    ref = orc.render(oracle_scene(a), u, oracle_params(p), 1, 1, 1, taps=True, n_threads=8)
    bad = got["tri"] != ref["tri"]
    print(pos, "hit agree", 1 - bad.mean(), "oracle prims at bad", np.bincount(ref["prim"][bad] + 1, minlength=9), "gpu prims at bad", np.bincount(got["prim"][bad] + 1, minlength=9))
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:4]:
        print("   pixel", x, y, "oracle", ref["tri"][y, x], ref["prim"][y, x], ref["t"][y, x], "gpu", got["tri"][y, x], got["prim"][y, x], got["t"][y, x])
ctx.close()
