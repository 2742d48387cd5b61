#!/usr/bin/env python3
"""Latency of the extend stage's rounds when a wave is alone on the GPU: n rays (one wave for
n <= 64) through rthip_trace_rays; extend time / rounds executed.
    python tools/round_latency.py [bench.py workload arguments]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse_args(sys.argv[1:])
    from raytracer_amd import device
    scene, name = bench.build_workload(args)
    arrays = scene.arrays()
    ctx = device.Context(0)
    ctx.upload_scene(arrays)
    g = scene.camera(0).to_gpu()
    o0 = np.array(g["origin"], np.float32)
    d0 = np.array(g["dir"], np.float32)
    rng = np.random.default_rng(0)
    for n in (1, 8, 64, 256, 4096):
        o = np.tile(o0, (n, 1))
        d = d0 + 0.25 * rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pt = device.default_params(stack_depth=args.stack_depth, flags=device.FLAG_TIMING)
        pc = device.default_params(stack_depth=args.stack_depth, flags=device.FLAG_COUNTERS)
        ctx.trace_rays(o, d, None, pt)
        ms = []
        for _ in range(5):
            ctx.reset_timing()
            ctx.trace_rays(o, d, None, pt)
            ms.append(ctx.timing()["extend"]["ms"])
        ctx.reset_counters()
        ctx.trace_rays(o, d, None, pc)
        c = ctx.counters()
        rounds = c["rounds_prim"] + c["rounds_inner"] + c["rounds_leaf"]
        steps = c["prims_visited"] + c["inner_steps"] + c["leaf_steps"]
        print(json.dumps({"rays": n, "extend_us": round(min(ms) * 1e3, 1), "rounds": rounds, "lane_steps": steps,
                          "us_per_round": round(min(ms) * 1e3 / max(rounds, 1), 3),
                          "rounds_inner": c["rounds_inner"], "rounds_leaf": c["rounds_leaf"], "rounds_prim": c["rounds_prim"]}))
    ctx.close()


if __name__ == "__main__":
    main()
