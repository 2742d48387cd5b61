#!/usr/bin/env python3
"""Where one 1-spp frame per call (what the reference's frame() does, main.rs:356-407) spends its
time: wall time per call and HIP-event time per stage, for the headline workload.
    python tools/single_frame.py [--frames 1] [--calls 20] [--staged] [bench.py workload arguments]
--staged sends the call through the staged pipeline (RTHIP_FLAG_STAGED) instead of the single launch
that small frame groups get."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    argv = sys.argv[1:]
    frames, calls = 1, 20
    staged = "--staged" in argv
    if staged:
        argv.remove("--staged")
    for name in ("--frames", "--calls"):
        if name in argv:
            i = argv.index(name)
            v = int(argv[i + 1])
            del argv[i:i + 2]
            if name == "--frames":
                frames = v
            else:
                calls = v
    args = bench.parse_args(argv)
    from raytracer_amd import device
    scene, name = bench.build_workload(args)
    arrays = scene.arrays()
    res = tuple(int(v) for v in args.res.lower().split("x"))
    ctx = device.Context(0)
    ctx.upload_scene(arrays)
    ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
    base = dict(max_bounces=args.max_bounces, term_prob=args.term_prob, stack_depth=args.stack_depth)
    extra = device.FLAG_STAGED if staged else 0
    bare = device.default_params(flags=extra, **base)
    timed = device.default_params(flags=device.FLAG_TIMING | extra, **base)
    for f in range(3):
        ctx.render(1 + f * frames, frames, 1, bare)
    ctx.synchronize()
    t0 = time.perf_counter()
    for f in range(calls):
        ctx.render(100 + f * frames, frames, 1, bare)
        ctx.synchronize()
    wall = (time.perf_counter() - t0) / calls * 1e3
    ctx.reset_timing()
    for f in range(calls):
        ctx.render(100 + f * frames, frames, 1, timed)
    t = ctx.timing()
    out = {"workload": name, "frames_per_call": frames, "staged": staged, "wall_ms_per_call": round(wall, 4),
           "stage_ms_per_call": {s: round(v["ms"] / calls, 4) for s, v in t.items() if v["launches"]},
           "stage_sum_ms": round(sum(v["ms"] for v in t.values()) / calls, 4)}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
