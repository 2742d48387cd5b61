#!/usr/bin/env python3
"""One frame per call while the camera turns: every call discards the image (reject_hist = 1, as
frame() does for a camera that moved, main.rs:376-378).  Wall time per call, for a turn of
--degrees per frame about the vertical axis.
    python tools/moving_camera.py [--degrees 0.5] [--calls 40] [bench.py workload arguments]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    argv = sys.argv[1:]
    degrees, calls = 0.5, 40
    for name in ("--degrees", "--calls"):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            if name == "--degrees":
                degrees = float(v)
            else:
                calls = int(v)
    args = bench.parse_args(argv)
    from raytracer_amd import device
    scene, name = bench.build_workload(args)
    res = tuple(int(v) for v in args.res.lower().split("x"))
    ctx = device.Context(0)
    ctx.upload_scene(scene.arrays())
    base = np.ascontiguousarray(scene.frame_uniforms(res, frame=1, reject_hist=1)).view(np.uint8).reshape(-1).copy()
    p = device.default_params(max_bounces=args.max_bounces, term_prob=args.term_prob, stack_depth=args.stack_depth)

    def uniforms(k):
        u = base.copy()
        cam = u[512:560].view(np.float32)          # GpuCamera: dir, fovy, origin, focus, ...
        a = np.radians(degrees) * k
        x, y = float(cam[0]), float(cam[1])
        cam[0] = np.float32(x * np.cos(a) - y * np.sin(a))
        cam[1] = np.float32(x * np.sin(a) + y * np.cos(a))
        return u

    for k in range(3):
        ctx.set_frame_uniforms(uniforms(k))
        ctx.render(1 + k, 1, 1, p)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(calls):
        ctx.set_frame_uniforms(uniforms(3 + k))
        ctx.render(4 + k, 1, 1, p)
        ctx.synchronize()
    wall = (time.perf_counter() - t0) / calls * 1e3
    print(json.dumps({"workload": name, "degrees_per_frame": degrees, "wall_ms_per_call": round(wall, 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
