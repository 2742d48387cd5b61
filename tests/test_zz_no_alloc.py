"""rthip_trace_rays_noalloc (the entry point that allocates nothing) against rthip_trace_rays,
through the C ABI: bit-identical outputs on two calls in a row, the second with fewer rays, on a
scene whose BLEND materials advance the seeds, with the default and with a deep traversal stack."""
import numpy as np
import pytest

from raytracer_amd import device, procedural

pytestmark = pytest.mark.gpu


def rays_for(scene_arrays, n, seed):
    rng = np.random.default_rng(seed)
    v = scene_arrays["tris"]["v"][:, :, :3].reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    c, e = (lo + hi) / 2, (hi - lo)
    o = (c + rng.uniform(-0.8, 0.8, (n, 3)) * e).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    return o, d, seeds


def noalloc(ctx, torch, o, d, seeds, params):
    n = o.shape[0]
    dev = "cuda:0"
    rays = torch.from_numpy(np.ascontiguousarray(np.concatenate([o, d], axis=1))).to(dev)
    sd = torch.from_numpy(seeds.view(np.int32).copy()).to(dev)
    t = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
    tri = torch.full((n,), -7, dtype=torch.int32, device=dev)
    prim = torch.full((n,), -7, dtype=torch.int32, device=dev)
    ovf = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.trace_rays_noalloc(rays.data_ptr(), sd.data_ptr(), n, t.data_ptr(), tri.data_ptr(), prim.data_ptr(), ovf.data_ptr(), params)
    ctx.synchronize()
    return dict(t=t.cpu().numpy(), tri=tri.cpu().numpy(), prim=prim.cpu().numpy(), overflow=ovf.cpu().numpy(),
                seed=sd.cpu().numpy().view(np.uint32))


def compare(ctx, torch, arrays, n, seed, params):
    o, d, seeds = rays_for(arrays, n, seed)
    ref = ctx.trace_rays(o, d, seeds, params)
    got = noalloc(ctx, torch, o, d, seeds, params)
    for k in ("tri", "prim", "overflow", "seed"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["t"].tobytes() == ref["t"].tobytes()
    return ref


@pytest.mark.parametrize("scene_name", ["materials", "terrain_deep", "instanced", "instanced_pyramid", "instanced_lists"])
def test_noalloc_equals_trace_rays_twice_in_a_row(scene_name):
    import torch
    if scene_name == "materials":
        s, params, sizes = procedural.material_test_scene(tex=32, env_size=(64, 32)), device.default_params(), (70001, 3000)
    elif scene_name == "terrain_deep":  # stack_depth above the LDS levels: the deep columns of the work array
        s, params, sizes = procedural.terrain_scene(tiles=6, env_size=(64, 32)), device.default_params(stack_depth=40), (50000, 777)
    elif scene_name == "instanced":     # more than 128 primitives: the in-kernel candidate scan
        s, params, sizes = procedural.terrain_scene(tiles=10, n=12, env_size=(64, 32), instanced=True), device.default_params(), (device.NOALLOC_MAX_RAYS, 64)
    else:                               # a long primitive list: the walk of the bounds pyramid, with a deep stack as well;
                                        # in shuffled order renders and rthip_trace_rays use per-ray lists, the entry point without allocations the scan
        s, params, sizes = (procedural.terrain_scene(tiles=18, n=8, env_size=(64, 32), instanced=True, shuffled=scene_name.endswith("lists")),
                            device.default_params(stack_depth=40), (90000, 1000))
    a = s.arrays()
    ctx = device.Context(0)
    try:
        ctx.upload_scene(a)
        assert ctx.candidate_walk() == {"instanced": "scan", "instanced_pyramid": "pyramid", "instanced_lists": "list"}.get(scene_name, "mask")
        first = compare(ctx, torch, a, sizes[0], 1, params)
        second = compare(ctx, torch, a, sizes[1], 2, params)   # fewer rays: the work array still holds the first call's
        assert (first["tri"] >= 0).any() and (second["tri"] >= 0).any()
        if scene_name == "materials":
            sd = rays_for(a, sizes[0], 1)[2]
            assert (first["seed"] != sd).any()                  # BLEND alpha tests drew from the seeds
    finally:
        ctx.close()


def test_noalloc_rejects_more_rays_than_its_work_array_holds():
    s = procedural.box_scene(env_size=(64, 32))
    ctx = device.Context(0)
    try:
        ctx.upload_scene(s.arrays())
        with pytest.raises(device.RtHipError, match="exceed the fixed work memory"):
            ctx.trace_rays_noalloc(16, 0, device.NOALLOC_MAX_RAYS + 1, 0, 0, 0, 0)
    finally:
        ctx.close()


def test_noalloc_allocates_nothing_on_its_very_first_call():
    """A fresh context, a stack deeper than the LDS levels (the deep columns and the host-side walk
    for NaN rays are first used here), and the entry point called BEFORE rthip_trace_rays ever ran:
    the device's free memory is the same before and after, and the result still equals rthip_trace_rays'."""
    import torch
    s = procedural.terrain_scene(tiles=4, env_size=(64, 32))
    a = s.arrays()
    params = device.default_params(stack_depth=40)
    o, d, seeds = rays_for(a, 20000, 5)
    d[:8] = np.nan                      # NaN directions take the precomputed walk
    ctx = device.Context(0)
    try:
        ctx.upload_scene(a)
        dev = "cuda:0"
        rays = torch.from_numpy(np.ascontiguousarray(np.concatenate([o, d], axis=1))).to(dev)
        sd = torch.from_numpy(seeds.view(np.int32).copy()).to(dev)
        n = o.shape[0]
        t = torch.zeros(n, dtype=torch.float32, device=dev)
        tri = torch.zeros(n, dtype=torch.int32, device=dev)
        prim = torch.zeros(n, dtype=torch.int32, device=dev)
        ovf = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        ctx.trace_rays_noalloc(rays.data_ptr(), sd.data_ptr(), n, t.data_ptr(), tri.data_ptr(), prim.data_ptr(), ovf.data_ptr(), params)
        ctx.synchronize()
        free_after = torch.cuda.mem_get_info(0)[0]
        assert free_after == free_before, (free_before, free_after)
        ref = ctx.trace_rays(o, d, seeds, params)
        assert np.array_equal(tri.cpu().numpy(), ref["tri"]) and t.cpu().numpy().tobytes() == ref["t"].tobytes()
        assert np.all(ref["tri"][:8] == -1)
    finally:
        ctx.close()


def test_noalloc_calls_of_two_contexts_on_two_streams_do_not_mix():
    """Both contexts share the GPU's one work array; the library orders their calls (one event per
    GPU), so interleaved calls on different streams each get their own result."""
    import torch
    s = procedural.material_test_scene(tex=32, env_size=(64, 32))
    a = s.arrays()
    params = device.default_params()
    c1, c2 = device.Context(0), device.Context(0)
    try:
        for c in (c1, c2):
            c.upload_scene(a)
        sets = [rays_for(a, 60000 + 7000 * k, 10 + k) for k in range(6)]
        refs = [c1.trace_rays(o, d, sd, params) for o, d, sd in sets]
        dev = "cuda:0"
        bufs = []
        for o, d, sd in sets:
            n = o.shape[0]
            bufs.append(dict(rays=torch.from_numpy(np.ascontiguousarray(np.concatenate([o, d], axis=1))).to(dev),
                             sd=torch.from_numpy(sd.view(np.int32).copy()).to(dev),
                             t=torch.zeros(n, dtype=torch.float32, device=dev), tri=torch.zeros(n, dtype=torch.int32, device=dev),
                             prim=torch.zeros(n, dtype=torch.int32, device=dev), ovf=torch.zeros(n, dtype=torch.uint8, device=dev), n=n))
        torch.cuda.synchronize()
        for k, b in enumerate(bufs):     # alternate the contexts (each has a stream of its own), no host synchronisation in between
            c = (c1, c2)[k % 2]
            c.trace_rays_noalloc(b["rays"].data_ptr(), b["sd"].data_ptr(), b["n"], b["t"].data_ptr(), b["tri"].data_ptr(), b["prim"].data_ptr(), b["ovf"].data_ptr(), params)
        c1.synchronize()
        c2.synchronize()
        for b, ref in zip(bufs, refs):
            assert np.array_equal(b["tri"].cpu().numpy(), ref["tri"])
            assert b["t"].cpu().numpy().tobytes() == ref["t"].tobytes()
            assert np.array_equal(b["sd"].cpu().numpy().view(np.uint32), ref["seed"])
    finally:
        c1.close()
        c2.close()
