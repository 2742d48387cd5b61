#!/usr/bin/env python3
"""Incoherent-ray stress of the extend stage: N uniformly random rays over a workload's whole
extent through rthip_trace_rays, nothing else on the GPU, so that a rocprofv3 counter pass of this
command sees only that traffic.  Takes bench.py's workload arguments:

    python tools/random_rays.py --workload terrain --tiles 78 --stack-depth 48 --rays 16000000

Prints one JSON line shaped like bench.py's as far as tools/prof_summary.py reads it (a "step" is
one full-size launch)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    rays = 16_000_000
    argv = sys.argv[1:]
    if "--rays" in argv:
        i = argv.index("--rays")
        rays = int(argv[i + 1])
        del argv[i:i + 2]
    args = bench.parse_args(argv)
    from raytracer_amd import device
    scene, name = bench.build_workload(args)
    arrays = scene.arrays()
    ctx = device.Context(0)
    ctx.upload_scene(arrays)
    p = device.default_params(stack_depth=args.stack_depth)
    r = bench.random_rays_bench(ctx, arrays, rays, p, device)
    ctx.close()
    # launches of the timed instantiation: a 65536-ray warm-up (0.4 % of a full launch) and two full ones
    out = {"metric": "Mrays/sec (uniformly random rays)", "value": r["mrays_per_s"], "unit": "Mrays/s", "steps": 2, "warmup": 0,
           "rays_per_step_per_gpu": rays + 32768, "config": {"workload": name + ", uniformly random rays"}, "random_rays": r}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
