#!/usr/bin/env python3
"""Soak run: one context through many scene uploads, resolutions, frame-group sizes and ray queries in
a row; free device memory must come back to where it was once the context is closed, and must not
creep while it lives.    python tests/tools/soak_gpu.py [--rounds 40]
tests/test_gpu_soak.py runs a short form of it (soak(rounds=8)) under pytest."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ["RTHIP_TUNING"] = "1"   # the forced candidate walks are experiment knobs
import torch  # noqa: E402  (before the device library opens the GPU)
from raytracer_amd import device, procedural  # noqa: E402


def soak(n_rounds, log=print):
    """Two contexts one after the other, n_rounds each.  Returns the free device memory (bytes) before,
    the lowest seen while each context lived, and after each was closed."""
    rng = np.random.default_rng(1)
    torch.cuda.init()
    free0 = torch.cuda.mem_get_info()[0]
    scenes = [procedural.box_scene(env_size=(64, 32)), procedural.material_test_scene(tex=32, env_size=(64, 32)),
              procedural.terrain_scene(tiles=12, n=8, env_size=(64, 32), instanced=True),
              procedural.terrain_scene(tiles=20, n=8, env_size=(64, 32), instanced=True),
              procedural.terrain_scene(tiles=20, n=8, env_size=(64, 32), instanced=True, shuffled=True),
              procedural.terrain_scene(tiles=70, n=4, env_size=(64, 32), instanced=True, shuffled=True)]
    arrays = [s.arrays() for s in scenes]
    def phase(rounds):
        c = device.Context(0)
        seen = []
        try:
            for r in range(rounds):
                k = int(rng.integers(0, len(scenes)))
                c.upload_scene(arrays[k])
                res = (int(rng.integers(8, 700)), int(rng.integers(8, 400)))
                c.set_frame_uniforms(scenes[k].frame_uniforms(res, frame=1, reject_hist=1))
                p = device.default_params(max_bounces=int(rng.integers(1, 6)), stack_depth=int(rng.choice([8, 23, 40])), enable_sun=int(rng.integers(0, 2)))
                c.render(1, int(rng.integers(1, 40)), 1, p)
                acc = c.read_accum()
                assert np.isfinite(acc).all() and acc[..., 3].min() >= 1.0
                o = rng.uniform(-50, 50, (1000, 3)).astype(np.float32)
                d = rng.normal(size=(1000, 3)).astype(np.float32)
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                c.trace_rays(o, d)
                c.resolve()
                seen.append(torch.cuda.mem_get_info()[0])
                log(f"round {r}: scene {k} ({arrays[k]['prims'].shape[0]} prims, {c.candidate_walk()}) res {res}: free {seen[-1] / 2**20:.0f} MiB")
        finally:
            c.close()
        return min(seen)

    # the first context also pays for what the runtime keeps for the process (code objects, the library's
    # fixed work array, kernel scratch): only the second one shows what the library itself leaves behind
    low1 = phase(n_rounds)
    free1 = torch.cuda.mem_get_info()[0]
    low2 = phase(n_rounds)
    free2 = torch.cuda.mem_get_info()[0]
    return free0, low1, free1, low2, free2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    args = ap.parse_args()
    free0, low1, free1, low2, free2 = soak(args.rounds, log=lambda m: print(m, flush=True))
    print(f"free before {free0 / 2**20:.0f} MiB; first context: lowest {low1 / 2**20:.0f}, after close {free1 / 2**20:.0f}; "
          f"second context: lowest {low2 / 2**20:.0f}, after close {free2 / 2**20:.0f} MiB")
    return 0 if free1 - free2 < (8 << 20) else 1


if __name__ == "__main__":
    sys.exit(main())
