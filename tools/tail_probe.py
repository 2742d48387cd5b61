#!/usr/bin/env python3
"""Diagnostic: where the time of one extend launch goes when a call renders few frames.
Builds a probe variant of the device library (-DRT_PROBE: every wave of k_extend records when it
started, when it first found the queue empty and when it ended, plus the most traversal steps one
of its rays took), renders single frames with it and prints, per bounce: the launch span, the
time at which the first / median / last wave saw the queue empty, and the step counts.
    python tools/tail_probe.py [--frames 1] [bench.py workload arguments]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "raytracer_amd", "lib", "librthip_probe.so")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "raytracer_amd", "csrc", "rthip.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
                           "-DRT_PROBE", "-o", LIB, src])


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "build":
        build()
        return
    frames = 1
    if "--frames" in argv:
        i = argv.index("--frames")
        frames = int(argv[i + 1])
        del argv[i:i + 2]
    os.environ["RTHIP_LIBRARY"] = LIB
    probe_file = "/tmp/rthip_probe.bin"
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
    for f in range(3):
        ctx.render(1 + f * frames, frames, 1, p)
    ctx.synchronize()
    if os.path.exists(probe_file):
        os.remove(probe_file)
    os.environ["RTHIP_PROBE_FILE"] = probe_file
    ctx.render(100, frames, 1, p)
    ctx.synchronize()
    del os.environ["RTHIP_PROBE_FILE"]
    raw = np.fromfile(probe_file, dtype=np.uint64)
    pos = 0
    while pos < raw.size:
        b, n = int(raw[pos]), int(raw[pos + 1])
        rec = raw[pos + 2: pos + 2 + 8 * n].reshape(n, 8).astype(np.int64)
        pos += 2 + 8 * n
        rec = rec[rec[:, 2] > 0]
        rounds_tail = (rec[:, 3] >> 20) & 0xFFFFFF
        live = rec[:, 3] >> 44
        rec[:, 3] &= 0xFFFFF
        tail_us = (rec[:, 2] - rec[:, 1]) / 100.0
        slow = np.argsort(rec[:, 2])[-5:]
        print(json.dumps({"end_percentiles_us": {str(q): round(float(np.percentile(rec[:, 2] - rec[:, 0].min(), q)) / 100.0, 1) for q in (50, 90, 99, 99.9, 100)},
                          "slowest_abs": [dict(start=round(float(rec[i, 0] - rec[:, 0].min()) / 100, 1), drain=round(float(rec[i, 1] - rec[:, 0].min()) / 100, 1),
                                               end=round(float(rec[i, 2] - rec[:, 0].min()) / 100, 1)) for i in slow]}))
        kinds = {}
        for j, kname in enumerate(("inner", "leaf", "prim")):
            cnt = rec[slow, 4 + j] & 0xFFFFFFFF
            tk = rec[slow, 4 + j] >> 32
            kinds[kname] = [dict(rounds=int(c), us_per_round=round(float(t) / 100.0 / max(int(c), 1), 3)) for c, t in zip(cnt, tk)]
        print(json.dumps({"slowest_waves_tail_rounds_by_kind": kinds}))
        print(json.dumps({"slowest_waves": [dict(tail_us=round(float(tail_us[i]), 1), rounds=int(rounds_tail[i]), live_at_drain=int(live[i]),
                                                 max_steps=int(rec[i, 3]), us_per_round=round(float(tail_us[i]) / max(int(rounds_tail[i]), 1), 3)) for i in slow],
                          "median_wave": dict(tail_us=round(float(np.median(tail_us)), 1), rounds=int(np.median(rounds_tail)), live_at_drain=int(np.median(live)))}))
        t0 = rec[:, 0].min()
        us = lambda t: round(float(t - t0) / 100.0, 1)   # 100 MHz ticks
        drain = rec[:, 1][rec[:, 1] > 0]
        end = rec[:, 2]
        print(json.dumps({"bounce": b, "waves": int(rec.shape[0]), "span_us": us(end.max()),
                          "last_wave_start_us": us(rec[:, 0].max()),
                          "queue_empty_first_us": us(drain.min()), "queue_empty_median_us": us(np.median(drain)),
                          "wave_end_median_us": us(np.median(end)), "wave_end_p99_us": us(np.percentile(end, 99)),
                          "max_steps_of_a_ray": int(rec[:, 3].max()), "median_of_wave_max_steps": int(np.median(rec[:, 3])),
                          "waves_alive_after_90pct_of_span": int((end > t0 + 0.9 * (end.max() - t0)).sum())}))
    ctx.close()


if __name__ == "__main__":
    main()
