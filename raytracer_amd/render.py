"""Headless render harness: what the reference's window loop does (main.rs:177-192 load a scene,
:356-407 one sample per frame(), present), without a window and with every literal of the
reference as a flag (model paths main.rs:178, resolution main.rs:464, bounce cap / termination
probability / aperture / stack depth shader.wgsl:898, 896, 765, 303).

    python -m raytracer_amd.render --scene model.glb --env sky.hdr --res 1920x1080 --spp 64 -o out.png
    python -m raytracer_amd.render --scene baked.rtscene --spp 1024 --gpus 8 -o out.png --dump-accum accum.npy

Frames first_frame .. first_frame + spp - 1 are rendered (the frame index is the RNG key,
shader.wgsl:895; the reference's first dispatch is frame 1, main.rs:372), accumulated, resolved
(mean, PBR Neutral, sRGB; fs_main shader.wgsl:994-1012 + the sRGB surface main.rs:116) and
written as an 8-bit PNG.  --gpus N stripes the frames over N GPUs of this node from this one
process (librthip_group) and sums the accumulators with one RCCL reduce.  There is no CPU path:
without the HIP library and a gfx950 device the command fails.
"""
import argparse
import json
import os
import struct
import sys
import time
import zlib

import numpy as np


def png_bytes(rgba8):
    """8-bit RGBA PNG of an (h, w, 4) uint8 array (filter 0 on every row)."""
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 4
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 4)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def parse_res(text):
    try:
        w, h = (int(v) for v in text.lower().split("x"))
    except ValueError:
        raise argparse.ArgumentTypeError(f"resolution {text!r} is not WIDTHxHEIGHT")
    if w <= 0 or h <= 0:
        raise argparse.ArgumentTypeError(f"resolution {text!r} is empty")
    return w, h


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m raytracer_amd.render", description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", required=True, help="glTF (.gltf/.glb) or baked .rtscene file")
    ap.add_argument("--env", default=None, help="Radiance .hdr environment map (2:1 equirectangular); default: a procedural sky")
    ap.add_argument("--res", type=parse_res, default=(512, 512), help="WIDTHxHEIGHT (the reference's window starts at 512x512, main.rs:464)")
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel = frames rendered")
    ap.add_argument("--first-frame", type=int, default=1)
    ap.add_argument("--max-bounces", type=int, default=8)
    ap.add_argument("--term-prob", type=float, default=0.5)
    ap.add_argument("--aperture", type=float, default=0.025)
    ap.add_argument("--stack-depth", type=int, default=23)
    ap.add_argument("--sun", action="store_true", help="the sun block the reference keeps commented out (shader.wgsl:916-937)")
    ap.add_argument("--camera", type=int, default=0, help="index of the scene camera (glTF cameras first; a default camera if the file has none)")
    ap.add_argument("--camera-blob", default=None, help="file with the 48 bytes of a GpuCamera (input.rs:20-32) to use instead, e.g. dumped by the reference's host")
    ap.add_argument("--focus", action="store_true", help="autofocus the camera on what it looks at (Scene::focus_camera, what a scene change does upstream, main.rs:319)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--linear", action="store_true", help="skip the sRGB transfer (the bytes fs_main writes before the surface encodes them)")
    ap.add_argument("-o", "--out", default=None, help="PNG to write")
    ap.add_argument("--dump-accum", default=None, help=".npy: float32 (h, w, 4) radiance sums + sample count")
    ap.add_argument("--dump-hits", default=None, help=".npz: primary-hit triangle / primitive / distance and final RNG state per pixel of the first frame")
    ap.add_argument("--dump-nodes", default=None, help=".npz: the device's node array and the device index of every uploaded node")
    return ap


def load_scene(args):
    from . import host, procedural
    if not os.path.exists(args.scene):
        raise SystemExit(f"render: scene file {args.scene} does not exist")
    if args.scene.lower().endswith(".rtscene"):
        scene = host.Scene.from_cache(args.scene)
        if args.env:
            scene.set_equirectangular_env_map(args.env)
    else:
        scene = host.Scene.from_path(args.scene, args.env)
    if scene.n_cameras == 0:
        scene.add_camera(host.Camera.default())
    a = scene.arrays()
    if a["env"].size == 0:
        scene.set_env(procedural.sky_env(1024, 512))
    if not (0 <= args.camera < scene.n_cameras):
        raise SystemExit(f"render: camera {args.camera} of {scene.n_cameras}")
    if args.focus:
        scene.focus_camera(args.camera)
    return scene


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.spp < 1 or args.gpus < 1:
        raise SystemExit("render: --spp and --gpus must be at least 1")
    if not (args.out or args.dump_accum or args.dump_hits or args.dump_nodes):
        raise SystemExit("render: nothing to write (give -o, --dump-accum, --dump-hits or --dump-nodes)")
    from . import device, layouts as L
    scene = load_scene(args)
    arrays = scene.arrays()
    cam = None
    if args.camera_blob:
        blob = np.fromfile(args.camera_blob, dtype=np.uint8)
        if blob.size != 48:
            raise SystemExit(f"render: {args.camera_blob} holds {blob.size} bytes, a GpuCamera has 48")
        cam = blob.view(L.GPU_CAMERA)[0]
    uniforms = scene.frame_uniforms(args.res, frame=args.first_frame, reject_hist=1, camera_idx=args.camera, gpu_camera=cam)
    flags = device.FLAG_COUNTERS | (device.FLAG_TAPS if args.dump_hits else 0)
    params = device.default_params(max_bounces=args.max_bounces, term_prob=args.term_prob, aperture_radius=args.aperture,
                                   stack_depth=args.stack_depth, enable_sun=1 if args.sun else 0, flags=flags)
    grp = None
    if args.gpus > 1:
        from . import group
        have = device.device_count()
        if have < args.gpus:
            raise SystemExit(f"render: --gpus {args.gpus} but this node has {have}")
        grp = group.Group(list(range(args.gpus)))
        grp.upload_scene(arrays)
        grp.set_frame_uniforms(uniforms)
        ctx = grp.contexts[0]
        for c in grp.contexts:
            c.reset_counters()
        t0 = time.perf_counter()
        grp.render(args.first_frame, args.spp, params)
        grp.reduce(0)
        dt = time.perf_counter() - t0
        rays = sum(c.counters()["rays"] for c in grp.contexts)
    else:
        ctx = device.Context(0)
        ctx.upload_scene(arrays)
        ctx.set_frame_uniforms(uniforms)
        ctx.reset_counters()
        t0 = time.perf_counter()
        ctx.render(args.first_frame, args.spp, 1, params)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        rays = ctx.counters()["rays"]
    if args.out:
        img = ctx.resolve(apply_srgb_oetf=not args.linear)
        with open(args.out, "wb") as f:
            f.write(png_bytes(img))
    if args.dump_accum:
        np.save(args.dump_accum, ctx.read_accum())
    if args.dump_hits:
        np.savez_compressed(args.dump_hits, **ctx.read_taps())
    if args.dump_nodes:
        np.savez_compressed(args.dump_nodes, nodes=ctx.read_nodes_device_layout(), device_index_of_node=ctx.read_node_index_map(),
                            tri_order=ctx.read_tri_order())
    print(json.dumps({"scene": os.path.basename(args.scene), "resolution": f"{args.res[0]}x{args.res[1]}", "spp": args.spp,
                      "first_frame": args.first_frame, "gpus": args.gpus, "triangles": int(arrays["tris"].shape[0]),
                      "primitives": int(arrays["prims"].shape[0]), "rays": int(rays), "seconds": round(dt, 4),
                      "mrays_per_s": round(rays / dt / 1e6, 1) if dt > 0 else None, "out": args.out}))
    if grp is not None:
        grp.close()
    else:
        ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
