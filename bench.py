#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path on the Sponza-shaped workload.

    python bench.py --gpus N --steps K --warmup W

A step is one 1-spp frame (generate + max_bounces x (cull + extend + shade) + accumulate) of the
1920x1080 workload.  N > 1 runs one rank per GPU (torch.distributed, RCCL): under
torch.distributed.run the ranks are given; started plainly, this script starts them itself as a
child launcher BEFORE touching any GPU, and refuses (exit status 2) when the node has fewer than N
GPUs or the launcher's world size is not N -- it never reports N GPUs from fewer devices.  Each
rank renders its own stripe of frame indices (rank r: frames r+1, r+1+N, ...) into a private
accumulator that one RCCL sum-reduce to rank 0 merges at the end (inside the timed region).
--spp-total S makes it a fixed job of S frames striped over the ranks (strong scaling).
--one-process drives the N GPUs from this one process through librthip_group instead.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, vendor figure
L2_PEAK_GBS = 34500.0   # aggregate L2 -> L1 rate of the eight XCDs
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_per_ray.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--spp-total", type=int, default=0,
                    help="strong scaling: a fixed job of this many frames striped over the GPUs (overrides --steps)")
    ap.add_argument("--one-process", action="store_true", help="drive all GPUs from this process (librthip_group) instead of one rank per GPU")
    ap.add_argument("--workload", default="atrium")
    ap.add_argument("--scene", default=None, help="path of a .rtscene file to render instead of a procedural workload")
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--max-bounces", type=int, default=4)
    ap.add_argument("--term-prob", type=float, default=0.5)
    ap.add_argument("--stack-depth", type=int, default=23)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--textures", default="sponza", choices=("sponza", "light"),
                    help="atrium: 'sponza' = an albedo, a metallic-roughness and a normal map per material (93 textures, 390 MB at --tex 1024); "
                         "'light' = the earlier dressing (10 shared textures, 36 MB)")
    ap.add_argument("--tiles", type=int, default=1)
    ap.add_argument("--terrain-n", type=int, default=60)
    ap.add_argument("--view", default="ground", choices=("ground", "top"), help="terrain camera: near the first tiles, or looking down on the whole tiling")
    ap.add_argument("--instanced", action="store_true", help="terrain tiles as instanced primitives instead of one merged BVH")
    ap.add_argument("--shuffled", action="store_true", help="instanced terrain: list the instances in random order")
    ap.add_argument("--enable-sun", action="store_true", help="the sun option (shadow rays); not the reference's live behaviour")
    ap.add_argument("--random-rays", type=int, default=0, help="also time this many uniformly random rays over the scene's extent through rthip_trace_rays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-case", action="store_true", help="skip the HBM-bound side measurement (random rays over a tiled terrain that does not fit the caches)")
    ap.add_argument("--hbm-tiles", type=int, default=48, help="the HBM case: terrain tiled this many times each way into one BVH (48: 18.8 M triangles, 1.5 GB of triangles and nodes)")
    ap.add_argument("--hbm-rays", type=int, default=8_000_000)
    ap.add_argument("--no-single-frame", action="store_true", help="skip the one-frame-per-call latency measurement")
    ap.add_argument("--opaque", action="store_true", help="diagnostic: treat MASK/BLEND materials as opaque")
    ap.add_argument("--first-frame", type=int, default=1, help="frame index of the first timed step (RNG key)")
    ap.add_argument("--pmc-key", default=None, help="file the run under this workload name in profiles/pmc_per_ray.json (e.g. when it is rendered from a baked --scene)")
    ap.add_argument("--save-scene", default=None, help="write the workload as a .rtscene file and exit (no GPU needed)")
    return ap.parse_args(argv)


def visible_gpus():
    """GPUs this process could use, counted without initialising any of them."""
    import torch
    return int(torch.cuda.device_count())


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def resolve_launch(args, env=os.environ):
    """How this invocation runs.  Returns ("rank", world, rank, local_rank) when it is one rank of
    a launcher's world, ("single",) for one GPU, ("group", n) for --one-process, or
    ("spawn", cmd) when it has to start the ranks itself.  Raises SystemExit(2) instead of ever
    running on fewer GPUs than --gpus names."""
    n = args.gpus
    if n < 1:
        raise SystemExit(2)
    in_launcher = "RANK" in env and "WORLD_SIZE" in env
    if in_launcher:
        world = int(env["WORLD_SIZE"])
        if world != n:
            print(f"bench.py: --gpus {n} but the launcher started WORLD_SIZE={world} ranks; refusing to report {n} GPUs", file=sys.stderr)
            raise SystemExit(2)
        if args.one_process:
            print("bench.py: --one-process cannot run under a process launcher", file=sys.stderr)
            raise SystemExit(2)
        return ("rank", world, int(env["RANK"]), int(env.get("LOCAL_RANK", env["RANK"])))
    if n == 1 and not args.one_process:
        return ("single",)
    have = visible_gpus()
    if have < n:
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to report {n} GPUs from fewer devices", file=sys.stderr)
        raise SystemExit(2)
    if args.one_process:
        return ("group", n)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return ("spawn", cmd)


def extend_algorithmic_bytes(c):
    """Bytes the extend stage must move for the work counted in `c`: 32 B per node record touched
    (a root per primitive entered, two children per inner step), 36 B per triangle test (nine
    floats) and 80 B per primitive entered (64 B of inverse transform + 16 B of root index, alpha
    settings and albedo reference)."""
    return (32 * (c["prims_visited"] + 2 * c["inner_steps"]) + 36 * c["tri_tests"] + 80 * c["prims_visited"])


def extend_lane_loads(c):
    """Vector loads the extend stage issues for the work counted in `c`, lane by lane: four per
    inner-node step (the sibling pair's record), three per triangle test, six per primitive entered
    (its record and the closest distance so far), three per ray fetched -- in 16-byte units: the
    12-byte loads of a triangle's rows count three quarters, which is what the pipeline charges
    them; and the wave-level loads that carried them (one per load of a step a wave executed)."""
    lane = 4 * c["inner_steps"] + 3 * 0.75 * c["tri_tests"] + 6 * c["prims_visited"] + 3 * c["rays"]
    wave = 4 * c["rounds_inner"] + 3 * 0.75 * c["rounds_leaf"] + 6 * c["rounds_prim"] + 3 * c["batches"]
    return lane, wave


def workload_key(args):
    """Name under which the hardware-counter figures of this workload are filed in profiles/pmc_per_ray.json."""
    if args.pmc_key:
        return args.pmc_key
    if args.scene:
        return "scene:" + os.path.basename(args.scene)
    k = f"{args.workload}-{args.res}-b{args.max_bounces}"
    if args.workload == "terrain":
        k += f"-t{args.tiles}-{args.view}" + ("-inst" if args.instanced else "") + ("-shuf" if args.shuffled else "")
    if args.workload == "atrium" and (args.detail != 1.0 or args.tex != 1024):
        k += f"-d{args.detail}-x{args.tex}"
    if args.workload == "atrium" and args.textures != "sponza":
        k += f"-{args.textures}"
    if args.term_prob != 0.5:
        k += f"-p{args.term_prob}"
    if args.opaque:
        k += "-opaque"
    if args.enable_sun:
        k += "-sun"
    return k


def pmc_figures(key):
    """Per-ray hardware-counter figures of k_extend for this workload, from the committed
    rocprofv3 passes (tools/prof.sh writes them); None when the workload has not been profiled."""
    try:
        return json.load(open(PMC_FILE)).get(key)
    except (OSError, ValueError):
        return None


def build_workload(args):
    from raytracer_amd import procedural
    if args.scene:
        # a baked scene (e.g. Khronos Sponza prepared elsewhere); its own first camera is used
        from raytracer_amd import host
        scene = host.Scene.from_cache(args.scene)
        if scene.n_cameras == 0:
            scene.add_camera(host.Camera.default())
        return scene, f"baked scene {os.path.basename(args.scene)}"
    if args.workload == "atrium":
        scene = procedural.atrium_scene(detail=args.detail, tex=args.tex, textures=args.textures)
        name = "sponza-proxy atrium (procedural; Khronos Sponza is not shipped with the reference)"
    elif args.workload == "blob":
        scene = procedural.blob_scene()
        name = "suzanne-proxy blob"
    elif args.workload == "terrain":
        scene = procedural.terrain_scene(tiles=args.tiles, n=args.terrain_n, instanced=args.instanced, view=args.view, shuffled=args.shuffled)
        name = f"terrain tiled {args.tiles}x{args.tiles}" + (" as instances" if args.instanced else "") + (" in shuffled order" if args.shuffled else "") + f", {args.view} view"
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    return scene, name


def granted_cores():
    """Host threads this process may use: the box's own setting when it has one (OMP_NUM_THREADS),
    else the CPUs in the process's affinity mask."""
    n = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if n > 0:
        return n
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(arrays, uniforms, params, target_seconds=12.0):
    """Times the CPU oracle (the build's own restatement, not the reference binary) on the first
    frames of the same workload, all host threads the box grants this process.  One untimed frame
    sizes the run to about target_seconds of CPU work."""
    import numpy as np
    from oracle import orc
    env = arrays["env"]
    sc = orc.make_scene(arrays["tris"], arrays["tri_exts"], arrays["nodes"], arrays["prims"],
                        arrays["texels"], env, env.shape[1], env.shape[0])
    op = orc.Params(params.max_bounces, params.term_prob, params.aperture_radius, params.stack_depth, params.enable_sun)
    cores = granted_cores()
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
            "lavapipe": "unavailable (no cargo and no Vulkan ICD on this box): the reference shader itself was not run; "
                        "this is the build's own C restatement of it",
            "sample": f"frames 1..{frames} of the same workload, all {h} rows ({r['counters']['rays']} rays, {dt:.1f} s)"}


def random_rays_bench(ctx, arrays, n, params, device):
    """n uniformly random rays over the scene's whole extent through rthip_trace_rays: nothing two
    neighbouring lanes touch is shared below the top of the tree.  Returns the extend stage's rate."""
    import numpy as np
    v = arrays["tris"]["v"][:, :, :3].reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    rng = np.random.default_rng(1)
    o = (lo + rng.random((n, 3), dtype=np.float32) * (hi - lo)).astype(np.float32)
    o[:, 2] = hi[2] + 1.0 + 50.0 * rng.random(n, dtype=np.float32)   # start above the terrain
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = device.default_params(stack_depth=params.stack_depth, flags=device.FLAG_TIMING)
    ctx.trace_rays(o[:65536], d[:65536], None, p)   # warm-up, sizes the staging
    ctx.trace_rays(o, d, None, p)
    ctx.reset_timing()
    ctx.reset_counters()
    pc = device.default_params(stack_depth=params.stack_depth, flags=device.FLAG_TIMING)
    q = ctx.trace_rays(o, d, None, pc)
    t = ctx.timing()["extend"]
    pcount = device.default_params(stack_depth=params.stack_depth, flags=device.FLAG_COUNTERS)
    ctx.trace_rays(o, d, None, pcount)
    c = ctx.counters()
    ms = t["ms"] / max(t["launches"], 1)
    return {"rays": n, "hit_fraction": round(float((q["tri"] >= 0).mean()), 4), "extend_ms": round(ms, 4),
            "mrays_per_s": round(n / (ms * 1e-3) / 1e6, 1) if ms > 0 else None,
            "algorithmic_gbs": round(extend_algorithmic_bytes(c) / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
            "inner_steps_per_ray": round(c["inner_steps"] / n, 2), "tri_tests_per_ray": round(c["tri_tests"] / n, 2)}


def hbm_case(ctx, args, device):
    """The HBM-bound case of the same kernel, measured in the same run: uniformly random rays over a
    tiled terrain merged into ONE BVH that is far larger than L2 + Infinity Cache.  Leaves the
    terrain uploaded in ctx."""
    from raytracer_amd import procedural
    t0 = time.perf_counter()
    scene = procedural.terrain_scene(tiles=args.hbm_tiles, n=args.terrain_n, view="top", env_size=(64, 32))
    arrays = scene.arrays()
    build_s = time.perf_counter() - t0
    ctx.upload_scene(arrays)
    p = device.default_params(stack_depth=48)
    r = random_rays_bench(ctx, arrays, args.hbm_rays, p, device)
    scene_bytes = int(arrays["tris"].nbytes + arrays["nodes"].nbytes)
    key = f"terrain-t{args.hbm_tiles}-random-rays"
    out = {"workload": f"terrain tiled {args.hbm_tiles}x{args.hbm_tiles} into one BVH, {args.hbm_rays} uniformly random rays (rthip_trace_rays), stack depth 48",
           "triangles": int(arrays["tris"].shape[0]), "nodes": int(arrays["nodes"].shape[0]), "triangle_and_node_bytes": scene_bytes,
           "host_build_s": round(build_s, 1), "rays": r["rays"], "extend_ms": r["extend_ms"], "mrays_per_s": r["mrays_per_s"],
           "inner_steps_per_ray": r["inner_steps_per_ray"], "tri_tests_per_ray": r["tri_tests_per_ray"],
           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           # the algorithmic bytes of this launch over its duration, as a fraction of the HBM peak
           "algorithmic_gbs": r["algorithmic_gbs"], "algorithmic_frac": round(r["algorithmic_gbs"] / HBM_PEAK_GBS, 4) if r["algorithmic_gbs"] else None,
           "pmc_workload": key, "traffic_gbs": None, "traffic_frac": None}
    pmc = pmc_figures(key)
    if pmc and r["extend_ms"]:
        gbs = (pmc["fetch_bytes_per_ray"] + pmc["write_bytes_per_ray"]) * r["rays"] / (r["extend_ms"] * 1e-3) / 1e9
        out.update({"traffic_gbs": round(gbs, 1), "traffic_frac": round(gbs / HBM_PEAK_GBS, 4), "l2_hit_rate": pmc.get("l2_hit_rate"),
                    "traffic_source": pmc["source"]})
    return out


def main():
    args = parse_args()
    if args.save_scene:
        scene, name = build_workload(args)
        scene.save(args.save_scene)
        print(f"wrote {args.save_scene}: {name}")
        return 0
    launch = resolve_launch(args)
    if launch[0] == "spawn":
        # the ranks are children of this process; nothing here has touched a GPU
        return subprocess.run(launch[1]).returncode

    # Libraries underneath write to file descriptor 1 on their own (RCCL prints a version banner when
    # its communicator is made): the run's standard output is kept for the one JSON line, everything
    # else that would go there goes to standard error.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    from raytracer_amd import device

    mode = launch[0]
    world, rank, local_rank = 1, 0, 0
    torch = dist = None
    grp = None
    if mode == "rank":
        _, world, rank, local_rank = launch
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(2)
    elif mode == "group":
        world = launch[1]

    res = tuple(int(v) for v in args.res.lower().split("x"))
    scene, workload_name = build_workload(args)
    arrays = scene.arrays()
    if args.opaque:
        arrays["prims"]["material"]["alpha_settings"] = 0
    uniforms = scene.frame_uniforms(res, frame=1, reject_hist=1)
    n_pix = res[0] * res[1]

    base = dict(max_bounces=args.max_bounces, term_prob=args.term_prob, stack_depth=args.stack_depth,
                enable_sun=1 if args.enable_sun else 0)
    p_plain = device.default_params(flags=device.FLAG_TIMING, **base)
    p_count = device.default_params(flags=device.FLAG_COUNTERS, **base)
    p_bare = device.default_params(**base)
    W = args.warmup
    strong = args.spp_total > 0
    total_frames = args.spp_total if strong else args.steps * world
    K = args.spp_total if strong else args.steps          # steps reported: whole-job frames (strong) or frames per GPU (weak)

    def stripe(r):  # frames of rank r: first, count (stride = world)
        first = args.first_frame + r
        count = (total_frames - r + world - 1) // world if total_frames > r else 0
        return first, count

    devices_seen = [0]
    if mode == "group":
        from raytracer_amd import group
        grp = group.Group(list(range(world)))
        if grp.size != args.gpus:
            raise SystemExit(2)
        devices_seen = list(range(grp.size))
        grp.upload_scene(arrays)
        grp.set_frame_uniforms(uniforms)
        ctx = grp.contexts[0]
        for c in grp.contexts:
            c.reserve(max(stripe(0)[1], 1))
        if W > 0:
            grp.render(args.first_frame, W * world, p_bare)
        grp.reduce(0)   # untimed: the first reduce creates the RCCL communicator
        grp.synchronize()
        for c in grp.contexts:
            c.reset_timing()
        grp.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
        grp.synchronize()
        t0 = time.perf_counter()
        grp.render(args.first_frame, total_frames, p_plain)
        grp.synchronize()
        t_render = time.perf_counter() - t0
        grp.reduce(0)                                  # synchronous: every GPU's stream is drained
        elapsed = time.perf_counter() - t0
        rank_ms = [[round(t_render * 1e3, 3), round((elapsed - t_render) * 1e3, 3)]] * world   # one process: the group's times
        timing = ctx.timing()
        for c in grp.contexts:
            c.reset_counters()
        grp.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
        grp.render(args.first_frame, total_frames, p_count)
        grp.synchronize()
        per_rank = [c.counters() for c in grp.contexts]
        counters = per_rank[0]
        rays_local = counters["rays"]
        rays_total = sum(c["rays"] for c in per_rank)
    else:
        ctx = device.Context(local_rank)
        ctx.upload_scene(arrays)
        ctx.set_frame_uniforms(uniforms)
        accum_t = None
        if mode == "rank":
            accum_t = torch.zeros(n_pix * 4, dtype=torch.float32, device=f"cuda:{local_rank}")
            ctx.bind_accum(accum_t.data_ptr(), n_pix)
            # which devices really take part: one entry per rank, gathered from every rank
            mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=f"cuda:{local_rank}")
            seen = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(seen, mine)
            devices_seen = sorted({int(t.item()) for t in seen})
        first, count = stripe(rank)
        ctx.reserve(max(count, 1))  # size the work buffers for the timed call now, not inside it
        if W > 0:
            ctx.render(first, W, world, p_bare)
        ctx.synchronize()
        if mode == "rank":  # untimed: the first collective on a communicator sets up its channels
            dist.reduce(accum_t, dst=0, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        ctx.reset_timing()
        ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))

        def sync_all():
            ctx.synchronize()
            if mode == "rank":
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()

        sync_all()
        t0 = time.perf_counter()
        ctx.render(first, count, world, p_plain)
        ctx.synchronize()
        t_render = time.perf_counter() - t0
        if mode == "rank":
            dist.reduce(accum_t, dst=0, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        rank_ms = [[round(t_render * 1e3, 3), round((elapsed - t_render) * 1e3, 3)]]
        if mode == "rank":
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            # what each rank spent rendering and (from the end of its own render to the end of the
            # reduce, i.e. including the wait for the slowest rank) reducing
            mine_ms = torch.tensor(rank_ms[0], dtype=torch.float64, device=f"cuda:{local_rank}")
            all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
            dist.all_gather(all_ms, mine_ms)
            rank_ms = [[round(float(v), 3) for v in t.tolist()] for t in all_ms]
        timing = ctx.timing()

        # identical frames again with counters on: the RNG is keyed by (pixel, frame), so the work
        # is the same work; counting is kept out of the timed region
        ctx.reset_counters()
        ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
        ctx.render(first, count, world, p_count)
        counters = ctx.counters()
        rays_local = counters["rays"]
        rays_total = rays_local
        if mode == "rank":
            rt = torch.tensor([rays_local], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(rt, op=dist.ReduceOp.SUM)
            rays_total = int(rt.item())

    n_gpus = len(devices_seen) if mode != "single" else 1
    if n_gpus != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the ranks ran on {n_gpus} distinct device(s): {devices_seen}", file=sys.stderr)
        raise SystemExit(2)

    if rank == 0:
        local_frames = max(stripe(0)[1], 1)
        copy_gbs = ctx.copy_bandwidth()
        ext = timing["extend"]
        ext_bytes = extend_algorithmic_bytes(counters)
        launches = max(ext["launches"], 1)
        avg_ms = ext["ms"] / launches
        rays_per_launch = rays_local / launches
        algorithmic_gbs = (ext_bytes / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        key = workload_key(args)
        pmc = pmc_figures(key)
        lane_loads, wave_loads = extend_lane_loads(counters)
        active = max(1, min(64, int(round(lane_loads / max(wave_loads, 1)))))
        # lane-loads / s, measured now, on this GPU, for four access shapes: one or two 64-byte records
        # (the inner-node step's) or 128-byte records (a 4-wide step with exact boxes) per lane and step
        shapes = {"64-byte records, one in flight": (64, 1), "64-byte records, two in flight": (64, 2),
                  "128-byte records, one in flight": (128, 1), "128-byte records, two in flight": (128, 2)}
        ceilings = {k: ctx.gather_ceiling(active, *v) for k, v in shapes.items()}
        ceiling = max(ceilings.values())
        lane_rate = (lane_loads / launches) / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        roof = {
            # A byte-moving f32 / integer kernel with no contraction: HBM is the roofline it is classed
            # under.  `achieved` = the algorithmic bytes of a launch (the counting rule of
            # extend_algorithmic_bytes on this run's own traversal counters) / the launch's duration,
            # measured here with HIP events on the library's stream; `frac` = achieved / the HBM peak.
            # Where the working set lives in L1 / L2 / Infinity Cache that fraction EXCEEDS 1 (the
            # caches deliver the bytes, HBM does not have to): HBM is then not what binds, `binding`
            # below names what does, and `hbm_case` is the same kernel where HBM does bind.
            "bound": "hbm",
            "kernel": "k_extend",
            "achieved": round(algorithmic_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(algorithmic_gbs / HBM_PEAK_GBS, 4),
            "frac_note": ("above 1: the scene's %.0f MB of nodes and triangles are served by the caches; see binding and hbm_case"
                          % ((arrays["tris"].nbytes + 2 * arrays["nodes"].nbytes) / 1e6)) if algorithmic_gbs > HBM_PEAK_GBS else None,
            "traffic": None,
            "peak_stream_copy": round(copy_gbs, 1),
            "launches": ext["launches"],
            "avg_launch_ms": round(avg_ms, 5),
            "rays_per_launch": round(rays_per_launch, 1),
            "algorithmic_bytes_per_launch": round(ext_bytes / launches, 1),
            "algorithmic_gbs": round(algorithmic_gbs, 2),
            "pmc_workload": key,
            # The ceiling that binds when the scene is cache-resident: the CU's vector-memory pipeline
            # (address unit, L1 tag look-up, data return).  Unit = lane-loads: one lane's 16-byte load
            # (a 12-byte load counts 3/4).  achieved = the loads the traversal counters imply (4 per inner
            # step, 3 x 3/4 per triangle test, 6 per primitive entered, 3 per ray fetched) per second of k_extend;
            # ceiling = the best of rthip_gather_ceiling's access shapes at the same number of active lanes
            # per wave-level load: records of the traversal's size from an L1-resident table, no
            # dependence between steps (the 64-byte shape with one record in flight, which is the inner
            # step's own, is reported beside it: the stage can pass that one, since its rounds overlap).
            "binding": {
                "resource": "CU vector-memory pipeline: lane-loads of the divergent record fetches (address unit, L1 look-up, data return)",
                "unit": "Glane-loads/s (16-byte loads)",
                "achieved": round(lane_rate / 1e9, 1),
                "ceiling": round(ceiling / 1e9, 1),
                "frac": round(lane_rate / ceiling, 4) if ceiling > 0 else None,
                "active_lanes_per_wave_load": active,
                "lane_loads_per_ray": round(lane_loads / max(rays_local, 1), 1),
                "ceiling_source": "rthip_gather_ceiling, this run, this GPU: the best of four access shapes out of the L1, 8 waves per SIMD, independent steps",
                "ceiling_by_shape": {k: round(v / 1e9, 1) for k, v in ceilings.items()},
                # a 4-wide tree with exact f32 boxes would fetch 128 bytes per step
                "ceiling_with_128_byte_records": round(ceilings["128-byte records, one in flight"] / 1e9, 1),
            },
        }
        if pmc and avg_ms > 0:
            fabric_bytes = (pmc["fetch_bytes_per_ray"] + pmc["write_bytes_per_ray"]) * rays_per_launch
            fabric_gbs = fabric_bytes / (avg_ms * 1e-3) / 1e9
            l2_gbs = pmc["l2_read_bytes_per_ray"] * rays_per_launch / (avg_ms * 1e-3) / 1e9
            roof.update({
                # what left the L2s towards Infinity Cache + HBM per launch, from the hardware counters
                # (FETCH_SIZE x 2 on gfx950, + WRITE_SIZE; separate rocprofv3 --pmc passes of this
                # command, per ray, scaled by this run's rays).  FETCH_SIZE counts requests that the
                # Infinity Cache then serves as well: an upper bound of the HBM bytes.
                "traffic": round(fabric_bytes, 1),
                "fabric_traffic_gbs": round(fabric_gbs, 1),
                "fabric_traffic_frac": round(fabric_gbs / HBM_PEAK_GBS, 4),
                "fabric_traffic_frac_of_stream_copy": round(fabric_gbs / copy_gbs, 4) if copy_gbs > 0 else None,
                "traffic_over_algorithmic": round(fabric_bytes / max(ext_bytes / launches, 1.0), 4),
                "traffic_source": pmc["source"],
                "traffic_profiled_at": pmc.get("bench_args"),
                "l2_hit_rate": pmc.get("l2_hit_rate"),
                "l2_to_l1_gbs": round(l2_gbs, 1),
                "l2_frac": round(l2_gbs / L2_PEAK_GBS, 4),
                # busy (non-idle) fractions of the CU's vector-memory units over the dispatch
                "measured_limiter": {"unit": pmc.get("limiter_unit"), "busy_frac": pmc.get("limiter_busy_frac"),
                                     "address_unit_busy": pmc.get("ta_busy"), "data_return_busy": pmc.get("td_busy"),
                                     "data_return_stalled_on_l1_miss": pmc.get("td_tc_stall"),
                                     "valu_lane_utilisation": pmc.get("valu_lane_utilisation")},
            })
        out = {
            "metric": ("Mrays/sec (Sponza 1080p, 4 bounces) -- measured on a procedural Sponza-proxy atrium"
                       if args.workload == "atrium" and not args.scene else f"Mrays/sec ({workload_name}, {args.max_bounces} bounces)"),
            "value": round(rays_total / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "ranks_seen": world,
            "devices_seen": devices_seen,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / max(local_frames if not strong else K, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload_name,
                "triangles": int(arrays["tris"].shape[0]),
                "primitives": int(arrays["prims"].shape[0]),
                "nodes": int(arrays["nodes"].shape[0]),
                "texture_mb": round(arrays["texels"].nbytes / 1e6, 1),
                "textures": int(len({(int(o), int(z)) for ref in ("albedo", "emissive", "normal", "metallic_roughness")
                                     for o, z in zip(arrays["prims"]["material"][ref]["offset"], arrays["prims"]["material"][ref]["size"]) if z})),
                "resolution": f"{res[0]}x{res[1]}",
                "spp_per_step": 1,
                "max_bounces": args.max_bounces,
                "term_prob": args.term_prob,
                "depth_of_field": True,
                "sun": bool(args.enable_sun),
                "triangle_layout": "48-byte (p0, e1, e2) records, three 12-byte loads per test (not 9-plane SoA)",
                "parallelism": (f"spp-split x{n_gpus}" + (" (one process, librthip_group)" if mode == "group" else "")) if n_gpus > 1 else "single",
                "frames_total": total_frames,
                "frames_in_flight": ctx.frames_in_flight(),
                "candidate_walk": ctx.candidate_walk(),
            },
            "rays_per_step_per_gpu": round(rays_local / local_frames, 1),
            # per rank: milliseconds in its own render, and from the end of that to the end of the reduce
            "render_ms_per_rank": [r[0] for r in rank_ms],
            "reduce_ms_per_rank": [r[1] for r in rank_ms],
            "stage_ms_per_step": {s: round(timing[s]["ms"] / local_frames, 4) for s in ("generate", "cull", "extend", "shade", "shadow", "accumulate")},
            "roofline": roof,
            "counters_per_ray": {k: round(counters[k] / max(rays_local, 1), 3)
                                 for k in ("prims_visited", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests")},
            "extend_simd_utilisation": {
                "prim_steps": round(counters["prims_visited"] / max(64 * counters["rounds_prim"], 1), 3),
                "inner_steps": round(counters["inner_steps"] / max(64 * counters["rounds_inner"], 1), 3),
                "tri_steps": round(counters["tri_tests"] / max(64 * counters["rounds_leaf"], 1), 3),
                "rounds_per_batch": round((counters["rounds_prim"] + counters["rounds_inner"] + counters["rounds_leaf"]) / max(counters["batches"], 1), 1),
            },
        }
        if mode == "single" and not args.no_single_frame:
            # what the reference's frame() does (main.rs:356-407): one sample per call, then present
            n_single = 20
            ctx.set_frame_uniforms(scene.frame_uniforms(res, frame=1, reject_hist=1))
            for f in range(3):
                ctx.render(args.first_frame + f, 1, 1, p_bare)
            ctx.synchronize()
            t0 = time.perf_counter()
            for f in range(n_single):
                ctx.render(args.first_frame + 3 + f, 1, 1, p_bare)
                ctx.synchronize()
            out["ms_per_frame_single"] = round((time.perf_counter() - t0) / n_single * 1e3, 4)
            t0 = time.perf_counter()
            for f in range(n_single):
                ctx.render(args.first_frame + 3 + n_single + f, 1, 1, p_bare)
                ctx.resolve()
            out["ms_per_frame_single_with_resolve_to_host"] = round((time.perf_counter() - t0) / n_single * 1e3, 4)
            # ... and with the image left on the device, where the reference's render pass leaves it (main.rs:395-401)
            # (device memory straight from the HIP runtime the library has loaded: importing torch after it brings a second runtime)
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            shown = ctypes.c_void_p()
            if hip.hipMalloc(ctypes.byref(shown), ctypes.c_size_t(res[0] * res[1] * 4)) == 0:
                t0 = time.perf_counter()
                for f in range(n_single):
                    ctx.render(args.first_frame + 3 + 2 * n_single + f, 1, 1, p_bare)
                    ctx.resolve_into(shown.value)
                out["ms_per_frame_single_with_resolve_on_device"] = round((time.perf_counter() - t0) / n_single * 1e3, 4)
                hip.hipFree(shown)
        if args.random_rays > 0 and mode == "single":
            out["random_rays"] = random_rays_bench(ctx, arrays, args.random_rays, p_plain, device)
        if mode == "single" and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arrays, uniforms, p_plain)
        if mode == "single" and not args.no_hbm_case:
            out["roofline"]["hbm_case"] = hbm_case(ctx, args, device)   # last: it replaces the scene in ctx
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if grp is not None:
        grp.close()
    else:
        ctx.close()
    if mode == "rank":
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
