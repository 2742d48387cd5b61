#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path on the Sponza-shaped workload.

    python bench.py --gpus N --steps K --warmup W

A step is one 1-spp frame (generate + max_bounces x (extend + shade)) of the 1920x1080
workload; N > 1 is launched by torch.distributed.run with one rank per GPU, each rank rendering
its own stripe of frame indices (rank r renders frames r+1, r+1+N, ...) into a private
accumulator that one RCCL sum-reduce to rank 0 merges at the end (inside the timed region).
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import device, procedural  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, vendor figure


def extend_algorithmic_bytes(c):
    """Bytes the extend stage must move for the work counted in `c`: 32 B per node record touched
    (a root per primitive entered, two children per inner step), 36 B per triangle test (nine
    floats) and 80 B per primitive entered (64 B of inverse transform + 16 B of root index, alpha
    settings and albedo reference)."""
    return (32 * (c["prims_visited"] + 2 * c["inner_steps"]) + 36 * c["tri_tests"] + 80 * c["prims_visited"])


PROFILE_SUMMARY = os.path.join(ROOT, "profiles", "r01_final_extend_summary.txt")


def profiled_extend_traffic():
    """HBM-side bytes per k_extend launch from the committed rocprofv3 PMC passes of this same
    default command (tools/prof.sh): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    tallies 128-byte requests at 64 bytes, so it is doubled.  None when the summary is missing."""
    try:
        text = open(PROFILE_SUMMARY).read()
    except OSError:
        return None
    block = text.split("\n  void k_extend<false, true, false>(ExtArgs)\n")
    if len(block) < 2:
        return None
    vals = {}
    for line in block[-1].split("\n\n")[0].splitlines():
        f = line.split()
        if len(f) >= 5 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[f[0]] = float(f[-1])
    if len(vals) != 2:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


PIPELINE_SUMMARY = os.path.join(ROOT, "profiles", "r01_final_extend_memory_pipeline.txt")


def profiled_extend_units():
    """Busy fractions of the CU's vector-memory units during k_extend, from the committed PMC pass
    of the default command (tools/prof_mem.sh): busy cycles summed over the 256 CUs / (cycles of the
    dispatch x 256).  GRBM_GUI_ACTIVE is summed over the 8 XCDs.  None when the summary is missing."""
    try:
        text = open(PIPELINE_SUMMARY).read()
    except OSError:
        return None
    block = text.split("\n  void k_extend<false, true, false>(ExtArgs)\n")
    if len(block) < 2:
        return None
    vals = {}
    for line in block[-1].split("\n\n")[0].splitlines():
        f = line.split()
        if len(f) >= 5 and f[0] in ("GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "TD_TC_STALL_sum"):
            vals[f[0]] = float(f[-1])
    if len(vals) != 4 or vals["GRBM_GUI_ACTIVE"] <= 0:
        return None
    cu_cycles = vals["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
    return {"address_unit_busy": round(vals["TA_TA_BUSY_sum"] / cu_cycles, 3),
            "data_return_busy": round(vals["TD_TD_BUSY_sum"] / cu_cycles, 3),
            "data_return_stalled_on_l1_miss": round(vals["TD_TC_STALL_sum"] / cu_cycles, 3),
            "source": "profiles/r01_final_extend_memory_pipeline.txt"}


def build_workload(args):
    if args.scene:
        # a baked scene (e.g. Khronos Sponza prepared elsewhere); its own first camera is used
        from raytracer_amd import host
        scene = host.Scene.from_cache(args.scene)
        if scene.n_cameras == 0:
            scene.add_camera(host.Camera.default())
        return scene, f"baked scene {os.path.basename(args.scene)}"
    if args.workload == "atrium":
        scene = procedural.atrium_scene(detail=args.detail, tex=args.tex)
        name = "sponza-proxy atrium (procedural; Khronos Sponza is not shipped with the reference)"
    elif args.workload == "blob":
        scene = procedural.blob_scene()
        name = "suzanne-proxy blob"
    elif args.workload == "terrain":
        scene = procedural.terrain_scene(tiles=args.tiles, n=args.terrain_n, instanced=args.instanced)
        name = f"terrain tiled {args.tiles}x{args.tiles}" + (" as instances" if args.instanced else "")
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    return scene, name


def cpu_baseline(arrays, uniforms, params, target_seconds=12.0):
    """Times the CPU oracle (the build's own restatement, not the reference binary) on the first
    frames of the same workload, all host threads the box grants this process.  One untimed frame
    sizes the run to about target_seconds of CPU work."""
    from oracle import orc
    env = arrays["env"]
    sc = orc.make_scene(arrays["tris"], arrays["tri_exts"], arrays["nodes"], arrays["prims"],
                        arrays["texels"], env, env.shape[1], env.shape[0])
    op = orc.Params(params.max_bounces, params.term_prob, params.aperture_radius, params.stack_depth, 0)
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    h = int(np.ascontiguousarray(uniforms).view(np.uint8).reshape(-1)[580:584].view(np.uint32)[0])
    t0 = time.perf_counter()
    orc.render(sc, uniforms, op, 1, 1, 1, n_threads=cores)   # one frame sizes the timed run
    frame_seconds = time.perf_counter() - t0
    frames = int(min(64, max(1, round(target_seconds / max(frame_seconds, 1e-3)))))
    t0 = time.perf_counter()
    r = orc.render(sc, uniforms, op, 1, frames, 1, n_threads=cores)
    dt = time.perf_counter() - t0
    rate = r["counters"]["rays"] / dt
    return {"value": round(rate / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"frames 1..{frames} of the same workload, all {h} rows ({r['counters']['rays']} rays, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="atrium")
    ap.add_argument("--scene", default=None, help="path of a .rtscene file to render instead of a procedural workload")
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--max-bounces", type=int, default=4)
    ap.add_argument("--term-prob", type=float, default=0.5)
    ap.add_argument("--stack-depth", type=int, default=23)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--tiles", type=int, default=1)
    ap.add_argument("--terrain-n", type=int, default=60)
    ap.add_argument("--instanced", action="store_true", help="terrain tiles as instanced primitives instead of one merged BVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opaque", action="store_true", help="diagnostic: treat MASK/BLEND materials as opaque")
    ap.add_argument("--first-frame", type=int, default=1, help="frame index of the first timed step (RNG key)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus and world > 1:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
    # under torch.distributed.run the collective path is used even for one rank, so that it can
    # be exercised on a one-GPU box
    distributed = world > 1 or "RANK" in os.environ

    torch = dist = None
    if distributed:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    res = tuple(int(v) for v in args.res.lower().split("x"))
    scene, workload_name = build_workload(args)
    arrays = scene.arrays()
    if args.opaque:
        arrays["prims"]["material"]["alpha_settings"] = 0
    uniforms = scene.frame_uniforms(res, frame=1, reject_hist=1)

    ctx = device.Context(local_rank if distributed else 0)
    ctx.upload_scene(arrays)
    ctx.set_frame_uniforms(uniforms)
    accum_t = None
    if distributed:
        accum_t = torch.zeros(res[0] * res[1] * 4, dtype=torch.float32, device=f"cuda:{local_rank}")
        ctx.bind_accum(accum_t.data_ptr(), res[0] * res[1])

    base = dict(max_bounces=args.max_bounces, term_prob=args.term_prob, stack_depth=args.stack_depth)
    p_plain = device.default_params(flags=device.FLAG_TIMING, **base)
    p_count = device.default_params(flags=device.FLAG_COUNTERS, **base)
    K, W = args.steps, args.warmup
    first = args.first_frame + rank
    stride = world

    ctx.reserve(K)  # size the work buffers for the timed call now, not inside it
    # warm-up (untimed)
    if W > 0:
        ctx.render(first, W, stride, device.default_params(**base))
    ctx.synchronize()
    ctx.reset_timing()
    ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))

    def sync_all():
        ctx.synchronize()
        if distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    ctx.render(first, K, stride, p_plain)
    ctx.synchronize()
    if distributed:
        dist.reduce(accum_t, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    timing = ctx.timing()

    # identical frames again with counters on: the RNG is keyed by (pixel, frame), so the work
    # is the same work; counting is kept out of the timed region
    ctx.reset_counters()
    ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
    ctx.render(first, K, stride, p_count)
    counters = ctx.counters()
    rays_local = counters["rays"]
    rays_total = rays_local
    if distributed:
        rt = torch.tensor([rays_local], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
        rays_total = int(rt.item())

    copy_gbs = ctx.copy_bandwidth() if rank == 0 else 0.0
    default_run = (args.scene is None and args.workload == "atrium" and args.res == "1920x1080" and args.max_bounces == 4 and args.detail == 1.0
                   and args.tex == 1024 and K % 64 == 0 and not args.opaque and os.environ.get("RTHIP_FRAMES_IN_FLIGHT") is None)
    if rank == 0:
        ext = timing["extend"]
        ext_bytes = extend_algorithmic_bytes(counters)
        launches = max(ext["launches"], 1)
        avg_ms = ext["ms"] / launches
        achieved = (ext_bytes / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "Mrays/sec (Sponza 1080p, 4 bounces)",
            "value": round(rays_total / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload_name,
                "triangles": int(arrays["tris"].shape[0]),
                "primitives": int(arrays["prims"].shape[0]),
                "nodes": int(arrays["nodes"].shape[0]),
                "texture_mb": round(arrays["texels"].nbytes / 1e6, 1),
                "resolution": f"{res[0]}x{res[1]}",
                "spp_per_step": 1,
                "max_bounces": args.max_bounces,
                "term_prob": args.term_prob,
                "depth_of_field": True,
                "parallelism": f"spp-split x{n_gpus}" if n_gpus > 1 else "single",
            },
            "rays_per_step_per_gpu": round(rays_local / K, 1),
            "stage_ms_per_step": {s: round(timing[s]["ms"] / K, 4) for s in ("generate", "cull", "extend", "shade", "accumulate")},
            "roofline": {
                "bound": "hbm",
                "kernel": "k_extend",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "peak_stream_copy": round(copy_gbs, 1),
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                # PMC counters cannot be read from inside the run; this is the figure of the committed
                # profile of the default command (same workload, same 64-frame launches)
                "traffic": profiled_extend_traffic() if default_run else None,
                "traffic_source": "profiles/r01_final_extend_summary.txt" if default_run else None,
                "launches": ext["launches"],
                "avg_launch_ms": round(avg_ms, 5),
                "algorithmic_bytes_per_launch": round(ext_bytes / launches, 1),
                # what the kernel is measured to be limited by: the CU's vector-memory front end,
                # not HBM (the algorithmic bytes are served by L1, L2 and the Infinity Cache)
                "measured_limiter": profiled_extend_units() if default_run else None,
            },
            "counters_per_ray": {k: round(counters[k] / max(rays_local, 1), 3)
                                 for k in ("prims_visited", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests")},
            "extend_simd_utilisation": {
                "prim_steps": round(counters["prims_visited"] / max(64 * counters["rounds_prim"], 1), 3),
                "inner_steps": round(counters["inner_steps"] / max(64 * counters["rounds_inner"], 1), 3),
                "tri_steps": round(counters["tri_tests"] / max(64 * counters["rounds_leaf"], 1), 3),
                "rounds_per_batch": round((counters["rounds_prim"] + counters["rounds_inner"] + counters["rounds_leaf"]) / max(counters["batches"], 1), 1),
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arrays, uniforms, p_plain)
        print(json.dumps(out), flush=True)

    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
