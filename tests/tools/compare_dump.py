#!/usr/bin/env python3
"""Pins this build against the reference shader itself, on a machine that can run it.

The reference's unmodified src/shader.wgsl runs headless through rust/ref-runner (wgpu on a
software adapter) and dumps its `screen` buffer; this tool prepares that runner's inputs and
compares its dump with the same frames rendered here.

    python tests/tools/compare_dump.py inputs  --scene model.glb [--env sky.hdr] --res 512x512 [--frame 1] -o prefix
        writes prefix.rtscene (the flat arrays upload_scene writes) and prefix.uniforms (608 bytes)
    python tests/tools/compare_dump.py compare --scene prefix.rtscene --uniforms prefix.uniforms --dump prefix.refdump [--renderer hip|oracle]
        renders the dump's frames with the HIP path (default when a GPU is present) or the CPU
        oracle and reports: equal sample counts, bit-identical pixels, PSNR of the resolved images
        (bar: >= 45 dB on converged images), and the reference's timing as Mrays/s
    python tests/tools/compare_dump.py make-dump --scene prefix.rtscene --uniforms prefix.uniforms --frames N -o file
        writes a dump in the runner's format from the CPU oracle (exercises the format without the runner)

Dump format (rust/ref-runner): "RTDUMP01", u32 w, h, frames, first_frame, f64 median ms per
frame, u32 threads, u32 0, then w*h vec4f.
"""
import argparse
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def read_dump(path):
    b = open(path, "rb").read()
    if b[:8] != b"RTDUMP01":
        raise SystemExit(f"{path} is not an RTDUMP01 file")
    w, h, frames, first = struct.unpack_from("<4I", b, 8)
    ms, = struct.unpack_from("<d", b, 24)
    threads, _ = struct.unpack_from("<2I", b, 32)
    screen = np.frombuffer(b, dtype=np.float32, count=w * h * 4, offset=40).reshape(h, w, 4).copy()
    return dict(w=w, h=h, frames=frames, first_frame=first, ms_per_frame=ms, threads=threads, screen=screen)


def write_dump(path, screen, frames, first_frame, ms=0.0, threads=0):
    h, w, _ = screen.shape
    with open(path, "wb") as f:
        f.write(b"RTDUMP01" + struct.pack("<4I", w, h, frames, first_frame) + struct.pack("<d", ms) + struct.pack("<2I", threads, 0))
        f.write(np.ascontiguousarray(screen, dtype=np.float32).tobytes())


def load(scene_path, uniforms_path):
    from raytracer_amd import host, layouts as L
    scene = host.Scene.from_cache(scene_path)
    u = np.fromfile(uniforms_path, dtype=np.uint8)
    if u.size != 608:
        raise SystemExit(f"{uniforms_path}: {u.size} bytes, FrameUniforms has 608")
    return scene, scene.arrays(), u.view(L.FRAME_UNIFORMS)[0]


def render(arrays, uniforms, first, frames, renderer):
    """(accum, rays) of frames first .. first + frames - 1, reference parameters."""
    if renderer == "hip":
        from raytracer_amd import device
        ctx = device.Context(0)
        ctx.upload_scene(arrays)
        ctx.set_frame_uniforms(uniforms)
        ctx.reset_counters()
        ctx.render(first, frames, 1, device.default_params(flags=device.FLAG_COUNTERS))
        out = ctx.read_accum(), ctx.counters()["rays"]
        ctx.close()
        return out
    from oracle import orc
    env = arrays["env"]
    sc = orc.make_scene(arrays["tris"], arrays["tri_exts"], arrays["nodes"], arrays["prims"], arrays["texels"], env, env.shape[1], env.shape[0])
    r = orc.render(sc, uniforms, first_frame=first, n_frames=frames, n_threads=min(os.cpu_count() or 1, 16))
    return r["accum"], r["counters"]["rays"]


def resolved(accum):
    """mean -> PBR Neutral -> sRGB -> [0,1] (fs_main + the sRGB surface), the space PSNR is quoted in."""
    from oracle import orc
    return orc.resolve(accum).astype(np.float64)[..., :3] / 255.0


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("inputs")
    a.add_argument("--scene", required=True)
    a.add_argument("--env", default=None)
    a.add_argument("--res", default="512x512")
    a.add_argument("--frame", type=int, default=1)
    a.add_argument("--focus", action="store_true")
    a.add_argument("-o", "--out", required=True)
    for name in ("compare", "make-dump"):
        c = sub.add_parser(name)
        c.add_argument("--scene", required=True)
        c.add_argument("--uniforms", required=True)
        if name == "compare":
            c.add_argument("--dump", required=True)
            c.add_argument("--renderer", choices=("hip", "oracle"), default=None)
        else:
            c.add_argument("--frames", type=int, default=1)
            c.add_argument("-o", "--out", required=True)
    args = ap.parse_args()

    if args.cmd == "inputs":
        from raytracer_amd import host, procedural
        if args.scene.lower().endswith(".rtscene"):
            scene = host.Scene.from_cache(args.scene)
            if args.env:
                scene.set_equirectangular_env_map(args.env)
        else:
            scene = host.Scene.from_path(args.scene, args.env)
        if scene.n_cameras == 0:
            scene.add_camera(host.Camera.default())
        if scene.arrays()["env"].size == 0:
            scene.set_env(procedural.sky_env(1024, 512))
        if args.focus:
            scene.focus_camera(0)
        w, h = (int(v) for v in args.res.lower().split("x"))
        scene.save(args.out + ".rtscene")
        np.ascontiguousarray(scene.frame_uniforms((w, h), frame=args.frame, reject_hist=1)).view(np.uint8).tofile(args.out + ".uniforms")
        print(f"wrote {args.out}.rtscene and {args.out}.uniforms ({w}x{h}, first frame {args.frame})")
        return 0

    scene, arrays, uniforms = load(args.scene, args.uniforms)
    if args.cmd == "make-dump":
        first = int(uniforms["frame"])
        accum, _ = render(arrays, uniforms, first, args.frames, "oracle")
        write_dump(args.out, accum, args.frames, first)
        print(f"wrote {args.out} from the CPU oracle")
        return 0

    d = read_dump(args.dump)
    if (d["w"], d["h"]) != tuple(int(v) for v in uniforms["res"]):
        raise SystemExit("the dump's resolution is not the uniforms' resolution")
    renderer = args.renderer
    if renderer is None:
        from raytracer_amd import device
        renderer = "hip" if device.device_count() > 0 else "oracle"
    accum, rays = render(arrays, uniforms, d["first_frame"], d["frames"], renderer)
    ref = d["screen"]
    same_count = bool(np.array_equal(ref[..., 3], accum[..., 3]))
    identical = float((ref.view(np.uint32) == accum.view(np.uint32)).all(axis=-1).mean())
    close = float(np.isclose(ref, accum, rtol=2e-3, atol=2e-4).all(axis=-1).mean())
    mse = float(np.mean((resolved(ref) - resolved(accum)) ** 2))
    psnr = 99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)
    out = {"renderer": renderer, "frames": d["frames"], "first_frame": d["first_frame"], "sample_counts_equal": same_count,
           "pixels_bit_identical": round(identical, 6), "pixels_within_tolerance": round(close, 6), "psnr_db": round(psnr, 2),
           "psnr_bar_db": 45.0, "max_abs_diff": float(np.abs(ref - accum).max()),
           "reference_ms_per_frame": d["ms_per_frame"], "reference_threads": d["threads"],
           "reference_mrays_per_s": round(rays / d["frames"] / (d["ms_per_frame"] * 1e-3) / 1e6, 4) if d["ms_per_frame"] > 0 else None}
    print(json.dumps(out))
    return 0 if same_count and (psnr >= 45.0 or d["frames"] < 16) else 1


if __name__ == "__main__":
    sys.exit(main())
