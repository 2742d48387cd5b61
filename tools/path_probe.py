#!/usr/bin/env python3
"""Diagnostic: the shape of the single launch that renders a small frame group (k_path).
Needs a probe build of the device library (bash tools/build_variant.sh probe -DRT_PROBE [knobs]):
every wave records when it started, when it found no camera path left and when it ended, its
rounds by kind and the lanes that had traversal work in them.
    python tools/path_probe.py [--frames 1] [--lib raytracer_amd/lib/variants/probe.so] [bench.py workload arguments]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    frames, lib = 1, os.path.join(ROOT, "raytracer_amd", "lib", "variants", "probe.so")
    for name in ("--frames", "--lib"):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            if name == "--frames":
                frames = int(v)
            else:
                lib = os.path.abspath(v)
    os.environ["RTHIP_LIBRARY"] = lib
    probe_file = "/tmp/rthip_path_probe.bin"
    import bench
    from raytracer_amd import device
    args = bench.parse_args(argv)
    scene, name = bench.build_workload(args)
    arrays = scene.arrays()
    res = tuple(int(v) for v in args.res.lower().split("x"))
    ctx = device.Context(0)
    ctx.upload_scene(arrays)
    ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
    p = device.default_params(max_bounces=args.max_bounces, term_prob=args.term_prob, stack_depth=args.stack_depth)
    for f in range(6):   # the order of the runs (longest first) comes from the call before
        ctx.render(1 + f * frames, frames, 1, p)
    ctx.synchronize()
    if os.path.exists(probe_file):
        os.remove(probe_file)
    os.environ["RTHIP_PROBE_FILE"] = probe_file
    ctx.render(100, frames, 1, p)
    ctx.synchronize()
    del os.environ["RTHIP_PROBE_FILE"]
    raw = np.fromfile(probe_file, dtype=np.uint64)
    n = int(raw[1])
    rec = raw[2:2 + 8 * n].reshape(n, 8).astype(np.int64)
    rec = rec[rec[:, 2] > 0]
    t0 = rec[:, 0].min()
    us = lambda t: round(float(t - t0) / 100.0, 1)   # 100 MHz ticks
    start, drain, end = rec[:, 0], rec[:, 1], rec[:, 2]
    trav, tail = rec[:, 3] & 0xFFFFFFFF, rec[:, 3] >> 32
    lanes, shade_ticks, lanes_tail, heavy_ticks = rec[:, 4] & ((1 << 40) - 1), (rec[:, 4] >> 40) << 4, rec[:, 5] & 0xFFFFFFFF, rec[:, 5] >> 32
    shade, shade_lanes = rec[:, 6] & 0xFFFFFFFF, rec[:, 6] >> 32
    scan, camera_ticks = rec[:, 7] & 0xFFFFFFFF, (rec[:, 7] >> 32) << 4
    print(json.dumps({
        "waves": int(rec.shape[0]), "span_us": us(end.max()), "last_wave_start_us": us(start.max()),
        "no_path_left_first_us": us(drain[drain > 0].min()), "no_path_left_median_us": us(np.median(drain[drain > 0])),
        "wave_end_us": {str(q): us(np.percentile(end, q)) for q in (10, 50, 90, 99, 100)},
        "rounds_per_wave_median": {"traversal": int(np.median(trav)), "after_drain": int(np.median(tail)), "path_rounds": int(np.median(shade)),
                                   "mask_scans": int(np.median(scan))},
        "rounds_total": {"traversal": int(trav.sum()), "after_drain": int(tail.sum()), "path_rounds": int(shade.sum()), "mask_scans": int(scan.sum())},
        "lanes_with_traversal_work_per_round": round(float(lanes.sum()) / max(int(trav.sum()), 1), 1),
        "the_same_after_the_drain": round(float(lanes_tail.sum()) / max(int(tail.sum()), 1), 1),
        "us_in_path_rounds_per_wave_median": round(float(np.median(heavy_ticks)) / 100.0, 1),
        "us_per_path_round": round(float(heavy_ticks.sum()) / 100.0 / max(int(shade.sum()), 1), 2),
        "of_which_shading_us": round(float(shade_ticks.sum()) / 100.0 / max(int(shade.sum()), 1), 2),
        "of_which_camera_paths_us": round(float(camera_ticks.sum()) / 100.0 / max(int(shade.sum()), 1), 2),
        "lanes_shaded_per_path_round": round(float(shade_lanes.sum()) / max(int(shade.sum()), 1), 1),
        "us_per_round_before_drain_median": round(float(np.median((drain - start)[drain > 0] / 100.0 / np.maximum((trav - tail + shade)[drain > 0], 1))), 3),
        "us_per_round_after_drain_median": round(float(np.median((end - drain)[drain > 0] / 100.0 / np.maximum(tail[drain > 0], 1))), 3),
    }))
    ctx.close()


if __name__ == "__main__":
    main()
