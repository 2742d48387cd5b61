"""Randomised differential cases of the HIP path against the CPU oracle, shared by the pytest
collection (tests/test_gpu_fuzz.py: a bounded, fixed-seed set) and the long-running tool
(tests/tools/fuzz_gpu.py).  A case = random scene kind, primitive order, candidate walk, resolution
(odd sizes too), bounce count, termination probability, stack depth, sun option, frame range, short
or staged form of the call; plus single rays through rthip_trace_rays, a third of them axis-parallel."""
import os

import numpy as np

from oracle import orc
from raytracer_amd import device, procedural
from util import oracle_params, oracle_scene, psnr, tonemapped

WALK_ENV = ("RTHIP_TREE_MIN_PRIMS", "RTHIP_CANDIDATE_WALK")


def make_case(rng):
    kind = str(rng.choice(["box", "materials", "terrain", "instanced", "instanced_shuffled", "atrium"]))
    walk = None
    if kind == "box":
        s = procedural.box_scene(env_size=(64, 32))
    elif kind == "materials":
        s = procedural.material_test_scene(tex=32, env_size=(64, 32))
    elif kind == "terrain":
        s = procedural.terrain_scene(tiles=int(rng.integers(1, 5)), n=int(rng.integers(6, 30)), env_size=(64, 32),
                                     view=str(rng.choice(["ground", "top"])))
    elif kind == "atrium":
        s = procedural.atrium_scene(detail=0.02, tex=32, env_size=(64, 32))
    else:
        tiles = int(rng.integers(9, 40))
        s = procedural.terrain_scene(tiles=tiles, n=int(rng.integers(4, 10)), env_size=(64, 32), instanced=True,
                                     shuffled=kind.endswith("shuffled"), view=str(rng.choice(["ground", "top"])))
        walk = str(rng.choice(["default", "scan", "pyramid", "list"]))
    res = (int(rng.integers(1, 200)), int(rng.integers(1, 120)))
    p = device.default_params(max_bounces=int(rng.integers(1, 9)), term_prob=float(rng.choice([0.5, 0.5, 0.25, 0.8])),
                              stack_depth=int(rng.choice([1, 3, 8, 13, 14, 23, 23, 40, 64])), enable_sun=int(rng.integers(0, 2)))
    first, n = int(rng.integers(1, 1000)), int(rng.integers(1, 5))
    staged = bool(rng.integers(0, 3) == 0)   # a third of the calls are sent through the staged pipeline (the others run as one launch: small images)
    return dict(kind=kind, walk=walk, scene=s, res=res, params=p, first=first, n=n, staged=staged)


def set_walk_env(walk, environ=os.environ):
    for k in WALK_ENV:
        environ.pop(k, None)
    if walk == "pyramid":
        environ["RTHIP_TREE_MIN_PRIMS"] = "8"
    elif walk == "scan":
        environ["RTHIP_TREE_MIN_PRIMS"] = "4000000000"
    elif walk == "list":
        environ["RTHIP_CANDIDATE_WALK"] = "list"


def run_case(case, rng, n_rays=48, threads=8):
    """Renders the case on the device and with the oracle.  Returns (failed, one-line report, walk used)."""
    s, res, p, first, n = case["scene"], case["res"], case["params"], case["first"], case["n"]
    a = s.arrays()
    set_walk_env(case["walk"])
    p.flags |= device.FLAG_COUNTERS | device.FLAG_TAPS
    if case["staged"]:
        p.flags |= device.FLAG_STAGED
    u = s.frame_uniforms(res, frame=first, reject_hist=1)
    c = device.Context(0)
    try:
        c.upload_scene(a)
        got_walk = c.candidate_walk()
        c.set_frame_uniforms(u)
        c.reset_counters()
        c.render(first, n, 1, p)
        got = dict(accum=c.read_accum(), counters=c.counters(), **c.read_taps())
        # single rays as well: random ones, and axis-parallel ones (zero direction components make the
        # reciprocal infinite: the slab tests then run on infinities, shader.wgsl:415-438)
        v = a["tris"]["v"][:, :, :3].reshape(-1, 3)
        tr = np.asarray(a["prims"]["transform"], np.float32).reshape(-1, 16)[:, 12:15]  # translations (column-major)
        lo, hi = tr.min(0) + v.min(0) - 1.0, tr.max(0) + v.max(0) + 1.0
        ro = rng.uniform(lo, hi, (n_rays, 3)).astype(np.float32)
        rd = rng.normal(size=(n_rays, 3)).astype(np.float32)
        rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        for k in range(0, n_rays, 3):
            ax = int(rng.integers(0, 3))
            rd[k] = 0.0
            rd[k, ax] = float(rng.choice([-1.0, 1.0]))
            if k % 2 == 0:
                ro[k] = np.round(ro[k])      # origins on round coordinates: on box faces and triangle planes of the procedural scenes
        pq = device.default_params(stack_depth=p.stack_depth)
        q = c.trace_rays(ro, rd, None, pq)
    finally:
        c.close()
        set_walk_env(None)
    sc_o = oracle_scene(a)
    op_q = oracle_params(pq)
    ray_bad = 0
    for k in range(n_rays):
        tri, prim, tt, _ = orc.trace_ray(sc_o, ro[k], rd[k], op_q, 0)
        if q["tri"][k] != tri or (tri >= 0 and (q["prim"][k] != prim or q["t"][k] != np.float32(tt))):
            ray_bad += 1
    ref = orc.render(sc_o, u, oracle_params(p), first, n, 1, taps=True, n_threads=threads)
    npx = res[0] * res[1]
    tri_ok = float((got["tri"] == ref["tri"]).mean())
    seed_ok = float((got["seed"] == ref["seed"]).mean())
    same = got["seed"] == ref["seed"]
    img_bad = int((~np.isclose(got["accum"], ref["accum"], rtol=2e-3, atol=2e-4).all(axis=-1) & same).sum())
    alpha_ok = np.array_equal(got["accum"][..., 3], ref["accum"][..., 3])
    cnt = {k: (got["counters"][k], ref["counters"][k]) for k in ("rays", "misses", "kills", "shade_calls", "stack_overflows", "shadow_rays")}
    cnt_bad = [k for k, (g, r) in cnt.items() if abs(g - r) > max(4, 4e-3 * r)]
    # Paths may part ways where an ulp of the device's sin / cos / pow decides a texel, a triangle edge or
    # a comparison several bounces later (textured scenes, long paths): a few per thousand at most.
    # The PSNR is reported for information: at 1-4 samples per pixel a handful of such pixels set it.
    db = psnr(tonemapped(got["accum"]), tonemapped(ref["accum"]))
    fail = bool((1.0 - tri_ok) * npx > max(2, 1e-3 * npx) or (1.0 - seed_ok) * npx > max(3, 4e-3 * npx) or img_bad > max(3, 3e-3 * npx)
                or not alpha_ok or cnt_bad or ray_bad)
    report = (f"{case['kind']} prims={a['prims'].shape[0]} walk={got_walk} res={res} bounces={p.max_bounces} p={p.term_prob} stack={p.stack_depth} "
              f"sun={p.enable_sun} frames={first}+{n} staged={case['staged']}: tri {tri_ok:.5f} seed {seed_ok:.5f} img_bad {img_bad} psnr {db:.1f} "
              f"alpha {alpha_ok} counters {cnt_bad or 'ok'} rays_bad {ray_bad}")
    return fail, report, got_walk
