"""Writes the small fixtures under tests/golden/ from the CPU oracle.

The reference has no tests or golden vectors and cannot be run here, so these fixtures pin the
ORACLE (regression: a later edit of the oracle or of the host builder that changes them is
flagged), not the reference.  The RNG vectors were derived by hand from shader.wgsl:641-659
with exact u32 arithmetic.  Run from the repo root:  python tests/tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orc  # noqa: E402
from raytracer_amd import procedural  # noqa: E402
from util import oracle_scene  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# integer RNG known answers (hand-derived, independent of the oracle)
rng = {
    "hash21": [[0, 0, 288866702], [1, 2, 428325059], [511, 511, 3707728273]],
    "pixel_streams": [
        {"pixel": [0, 0], "frame": 1, "seed0": 844223747,
         "draws": [[2309631892, 2005974744], [903460755, 2509538293], [3414464554, 1677724353], [1736744664, 1037676727]]},
        {"pixel": [1, 0], "frame": 1, "seed0": 857772177, "draws": [[1465755698, 2461887080]]},
        {"pixel": [255, 255], "frame": 1, "seed0": 1089891677, "draws": [[3094801689, 2248515535]]},
        {"pixel": [1919, 1079], "frame": 7, "seed0": 132983808, "draws": [[928494471, 4082706149]]},
    ],
}
json.dump(rng, open(os.path.join(OUT, "rng_kat.json"), "w"), indent=1)

# box scene (the stand-in for config 1): full node buffer, triangle order, 16x16 primary hits,
# 64-spp 16x16 accumulator
s = procedural.box_scene(env_size=(64, 32))
a = s.arrays()
res = (16, 16)
u = s.frame_uniforms(res, frame=1, reject_hist=1)
r1 = orc.render(oracle_scene(a), u, first_frame=1, n_frames=1, taps=True)
r64 = orc.render(oracle_scene(a), u, first_frame=1, n_frames=64)
np.savez_compressed(os.path.join(OUT, "box_scene.npz"), nodes=a["nodes"].view(np.uint8), tri_order=a["tri_order"],
                    tris=a["tris"].view(np.uint8), uniforms=np.frombuffer(u.tobytes(), np.uint8),
                    hit_tri=r1["tri"], hit_prim=r1["prim"], hit_t=r1["t"], seed=r1["seed"],
                    accum1=r1["accum"], accum64=r64["accum"])

# a closer camera so most pixels hit, all materials exercised
s = procedural.material_test_scene(tex=16, env_size=(32, 16))
a = s.arrays()
res = (24, 16)
u = s.frame_uniforms(res, frame=3, reject_hist=1)
r = orc.render(oracle_scene(a), u, first_frame=3, n_frames=2, frame_stride=5, taps=True)
np.savez_compressed(os.path.join(OUT, "materials_scene.npz"), nodes=a["nodes"].view(np.uint8), tri_order=a["tri_order"],
                    prims=a["prims"].view(np.uint8), uniforms=np.frombuffer(u.tobytes(), np.uint8),
                    hit_tri=r["tri"], hit_prim=r["prim"], hit_t=r["t"], seed=r["seed"], accum=r["accum"],
                    counters=np.array([r["counters"][k] for k in orc.COUNTER_NAMES], dtype=np.uint64))
print("wrote", sorted(os.listdir(OUT)))
