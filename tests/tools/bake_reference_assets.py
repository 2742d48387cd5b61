#!/usr/bin/env python3
"""Bakes the scenes that ship with the reference into fixtures under tests/golden/, so that the
GPU box -- which has no copy of the reference -- can run the HIP path on them.

    python tests/tools/bake_reference_assets.py [--reference-dir /root/reference]

For each of resources/Box.gltf, suzanne.glb, simple_terrain.glb, simple_with_camera.glb: the
flat arrays this build's host pipeline (librthost: glTF load, axis change, Bikker BVH build,
cameras) produces from the file, i.e. what the reference's scene.rs hands to upload_scene
(main.rs:328-353), as a gzip-compressed .rtscene file without environment map.  resources/
trail.hdr is kept once, as its decoded RGBE bytes (u8[512][1024][4]); rgba32f = mantissa *
2^(exponent - 136) reproduces the loader's floats exactly (checked below).  These are data derived
from the reference's assets (inputs of the hot path), not its source.

Also written: the oracle's primary-hit map of configuration 1 (Box.gltf, 512x512, frame 1, the
default camera of input.rs:202-219), which pins the oracle against later edits.
"""
import argparse
import gzip
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import host  # noqa: E402

ASSETS = ("Box.gltf", "suzanne.glb", "simple_terrain.glb", "simple_with_camera.glb")
OUT = os.path.join(ROOT, "tests", "golden")


def read_rgbe(path):
    """Radiance .hdr -> u8[h][w][4] RGBE (flat or new-style RLE scanlines, -Y h +X w)."""
    data = open(path, "rb").read()
    pos = data.index(b"\n\n") + 2
    end = data.index(b"\n", pos)
    f = data[pos:end].split()
    assert f[0] == b"-Y" and f[2] == b"+X", f
    h, w = int(f[1]), int(f[3])
    pos = end + 1
    out = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and data[pos] == 2 and data[pos + 1] == 2 and (data[pos + 2] << 8 | data[pos + 3]) == w:
            pos += 4
            for ch in range(4):
                x = 0
                while x < w:
                    n = data[pos]
                    if n > 128:
                        n -= 128
                        out[y, x:x + n, ch] = data[pos + 1]
                        pos += 2
                    else:
                        out[y, x:x + n, ch] = np.frombuffer(data, np.uint8, n, pos + 1)
                        pos += 1 + n
                    x += n
        else:
            out[y] = np.frombuffer(data, np.uint8, w * 4, pos).reshape(w, 4)
            pos += w * 4
    return out


def rgbe_to_float(rgbe):
    """The conversion of csrc/host/hdr.cpp: mantissa * 2^(e - 136), zero when e == 0, alpha 1."""
    e = rgbe[..., 3].astype(np.int32)
    out = np.ones(rgbe.shape[:2] + (4,), np.float32)
    out[..., :3] = np.ldexp(rgbe[..., :3].astype(np.float32), (e - 136)[..., None])
    out[..., :3][e == 0] = 0.0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference-dir", default="/root/reference")
    args = ap.parse_args()
    res_dir = os.path.join(args.reference_dir, "resources")
    os.makedirs(OUT, exist_ok=True)

    rgbe = read_rgbe(os.path.join(res_dir, "trail.hdr"))
    s = host.Scene()
    s.set_equirectangular_env_map(os.path.join(res_dir, "trail.hdr"))
    env = s.env_map_data
    assert env.shape == rgbe.shape and np.array_equal(env.view(np.uint32), rgbe_to_float(rgbe).view(np.uint32))
    np.savez_compressed(os.path.join(OUT, "ref_trail_rgbe.npz"), rgbe=rgbe)
    print(f"ref_trail_rgbe.npz: {rgbe.shape[1]}x{rgbe.shape[0]}")

    for name in ASSETS:
        scene = host.Scene.from_path(os.path.join(res_dir, name))
        stem = os.path.splitext(name)[0].lower()
        with tempfile.TemporaryDirectory() as tmp:
            raw = os.path.join(tmp, "s.rtscene")
            scene.save(raw)
            blob = open(raw, "rb").read()
        dst = os.path.join(OUT, f"ref_{stem}.rtscene.gz")
        with open(dst, "wb") as f:  # mtime 0: the same bytes on every run
            with gzip.GzipFile(filename="", mode="wb", fileobj=f, compresslevel=9, mtime=0) as g:
                g.write(blob)
        a = scene.arrays()
        print(f"{os.path.basename(dst)}: {a['tris'].shape[0]} triangles, {a['prims'].shape[0]} primitives, "
              f"{a['nodes'].shape[0]} nodes, {scene.n_cameras} cameras, {os.path.getsize(dst) / 1e6:.2f} MB")

    # configuration 1: the oracle's primary hits (pins the oracle, not the reference)
    from oracle import orc
    from util import load_reference_scene, oracle_scene
    scene, arrays = load_reference_scene("box")
    u = scene.frame_uniforms((512, 512), frame=1, reject_hist=1)
    r = orc.render(oracle_scene(arrays), u, first_frame=1, n_frames=1, taps=True, n_threads=8)
    np.savez_compressed(os.path.join(OUT, "ref_box_config1_taps.npz"), hit_tri=r["tri"].astype(np.int16),
                        hit_prim=r["prim"].astype(np.int8), rays=np.uint64(r["counters"]["rays"]))
    print("ref_box_config1_taps.npz:", int((r["tri"] >= 0).sum()), "primary hits,", r["counters"]["rays"], "rays")


if __name__ == "__main__":
    main()
