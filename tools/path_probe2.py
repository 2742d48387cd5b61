#!/usr/bin/env python3
"""Diagnostic (probe build with -DRT_PROBE -DRT_PROBE2): for every wave of the single launch, the path
that ended last in it: when (in wave rounds) it started and ended, and how many segments it had."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "raytracer_amd", "lib", "variants", "probe2.so")
os.environ["RTHIP_LIBRARY"] = lib
import bench
from raytracer_amd import device
args = bench.parse_args(sys.argv[1:])
scene, name = bench.build_workload(args)
arrays = scene.arrays()
res = tuple(int(v) for v in args.res.lower().split("x"))
ctx = device.Context(0)
ctx.upload_scene(arrays)
ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
p = device.default_params(max_bounces=args.max_bounces)
for f in range(6):
    ctx.render(1 + f, 1, 1, p)
ctx.synchronize()
pf = "/tmp/rthip_path_probe.bin"
if os.path.exists(pf):
    os.remove(pf)
os.environ["RTHIP_PROBE_FILE"] = pf
ctx.render(100, 1, 1, p)
ctx.synchronize()
raw = np.fromfile(pf, dtype=np.uint64)
n = int(raw[1])
rec = raw[2:2 + 8 * n].reshape(n, 8).astype(np.int64)
rec = rec[rec[:, 2] > 0]
t0 = rec[:, 0].min()
end_us = (rec[:, 2] - t0) / 100.0
drain_us = (rec[:, 1] - t0) / 100.0
st, en = rec[:, 6] & 0xFFFFFFFF, rec[:, 6] >> 32
b, run, iters = rec[:, 7] & 0xFF, (rec[:, 7] >> 8) & 0xFFFFFF, rec[:, 7] >> 32
order = np.argsort(-end_us)
print("span_us", end_us.max(), "drain median", np.median(drain_us))
for q in (0, 1, 2, 5, 10, 20, 40, 80, 160, 400, 1000, 2000):
    i = order[q]
    print(f"wave #{q} from the end: ends {end_us[i]:.0f} us, its last path: started at round {st[i]} of {iters[i]} ({st[i] / max(iters[i], 1):.2f}), "
          f"alive {en[i] - st[i]} rounds, bounce index at end {b[i]}, run {run[i]}")
late = order[:200]
print("last 200 waves: mean start fraction", float(np.mean(st[late] / np.maximum(iters[late], 1))), "mean final bounce", float(np.mean(b[late])),
      "mean alive rounds", float(np.mean(en[late] - st[late])))
print("all waves: mean start fraction", float(np.mean(st / np.maximum(iters, 1))), "mean final bounce", float(np.mean(b)), "mean alive rounds", float(np.mean(en - st)))
