#!/usr/bin/env python3
"""Do frame groups on several streams fill each other's launch tails?

    python tools/two_streams.py [--frames 20] [--parts 1,2,4] [--res 1920x1080]

The same F frames of the headline workload rendered (a) by one context in one group and (b) by P
contexts at once, each on its own stream, context p taking frames p+1, p+1+P, ... (the stripes of
the multi-GPU split, here on ONE GPU).  Prints the wall time of each form, best of a few repeats.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="20")
    ap.add_argument("--parts", default="1,2,4")
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--staged", action="store_true")
    args = ap.parse_args()
    from raytracer_amd import device, procedural
    res = tuple(int(v) for v in args.res.split("x"))
    scene = procedural.atrium_scene(tex=args.tex)
    arrays = scene.arrays()
    uniforms = scene.frame_uniforms(res, frame=1, reject_hist=1)
    parts = [int(v) for v in args.parts.split(",")]
    ctxs = []
    for _ in range(max(parts)):
        c = device.Context(0)
        c.upload_scene(arrays)
        c.set_frame_uniforms(uniforms)
        ctxs.append(c)
    p = device.default_params(max_bounces=4, flags=device.FLAG_STAGED if args.staged else 0)
    out = []
    for F in (int(v) for v in args.frames.split(",")):
        for P in parts:
            use = ctxs[:P]
            for c in use:
                c.reserve((F + P - 1) // P)
            best = None
            for rep in range(args.repeats + 1):
                for c in use:
                    c.set_frame_uniforms(uniforms)
                    c.synchronize()
                t0 = time.perf_counter()
                for k, c in enumerate(use):
                    n = (F - k + P - 1) // P
                    c.render(1 + k, n, P, p)
                for c in use:
                    c.synchronize()
                dt = (time.perf_counter() - t0) * 1e3
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            out.append({"frames": F, "contexts": P, "ms": round(best, 3), "ms_per_frame": round(best / F, 4)})
            print(json.dumps(out[-1]), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
