#!/usr/bin/env python3
"""Bake a glTF/GLB scene (+ optional Radiance environment map) into a .rtscene file.

    python tools/bake_scene.py scene.glb [env.hdr] -o scene.rtscene

The file holds the flat arrays the device library uploads (layout: csrc/host/cache.cpp) and is
what `bench.py --scene` and `Scene.from_cache` read.  PNG and JPEG textures are decoded here; the
reference's Rust host can write the same file (the layout is documented in csrc/host/cache.cpp).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from raytracer_amd import host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mesh")
    ap.add_argument("env", nargs="?", default=None)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--focus", action="store_true", help="autofocus camera 0 like Scene::from_path does on load")
    args = ap.parse_args()
    scene = host.Scene.from_path(args.mesh, args.env)
    if args.focus:
        scene.focus_camera(0)
    scene.save(args.out)
    a = scene.arrays()
    print(f"{args.out}: {a['tris'].shape[0]} triangles, {a['prims'].shape[0]} primitives, {a['nodes'].shape[0]} nodes, "
          f"{a['texels'].nbytes / 1e6:.1f} MB of texels, {scene.n_cameras} cameras")


if __name__ == "__main__":
    main()
