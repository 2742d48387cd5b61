"""Python face of the device library (include/rthip.h, librthip.so): one `Context` per GPU.

Plays the part of the reference's frame driver (main.rs:177-407): `upload_scene`, per-frame
uniforms, the compute pass (`render`) and the present pass (`resolve`).  There is no fallback:
without the HIP library or without a gfx950 device every call raises.
"""
import ctypes as C
import os

import numpy as np

from . import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RTHIP_LIBRARY") or os.path.join(_HERE, "lib", "librthip.so")

FLAG_COUNTERS, FLAG_TAPS, FLAG_TIMING, FLAG_STAGED = 1, 2, 4, 8
NOALLOC_MAX_RAYS = 1 << 18
STAGES = ("generate", "extend", "shade", "resolve", "accumulate", "shadow", "cull", "path")

EXPORTS = (
    "rthip_device_count", "rthip_create", "rthip_destroy", "rthip_last_error", "rthip_set_stream",
    "rthip_upload_scene", "rthip_set_frame_uniforms", "rthip_default_params", "rthip_render", "rthip_reserve",
    "rthip_bind_accum", "rthip_accum_device_ptr", "rthip_read_accum", "rthip_write_accum",
    "rthip_read_taps", "rthip_read_nodes_device_layout", "rthip_read_node_index_map",
    "rthip_read_tri_order", "rthip_read_tris_device_layout", "rthip_resolve",
    "rthip_get_counters", "rthip_reset_counters", "rthip_get_timing", "rthip_reset_timing",
    "rthip_synchronize", "rthip_copy_bandwidth", "rthip_trace_rays", "rthip_trace_rays_noalloc",
    "rthip_frames_in_flight", "rthip_candidate_walk", "rthip_gather_ceiling",
)


class RtHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rthip error {code}: {msg}")
        self.code = code


class SceneDesc(C.Structure):
    _fields_ = [
        ("tris", C.c_void_p), ("n_tris", C.c_uint64),
        ("tri_exts", C.c_void_p),
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
        ("prims", C.c_void_p), ("n_prims", C.c_uint32),
        ("texels", C.c_void_p), ("n_texels", C.c_uint64),
        ("env", C.c_void_p), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("max_bounces", C.c_uint32), ("term_prob", C.c_float), ("aperture_radius", C.c_float),
        ("stack_depth", C.c_uint32), ("enable_sun", C.c_uint32), ("flags", C.c_uint32),
    ]


COUNTER_NAMES = ("rays", "prims_visited", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests",
                 "texel_fetches", "shade_calls", "misses", "kills", "stack_overflows", "paths",
                 "rounds_prim", "rounds_inner", "rounds_leaf", "batches", "shadow_rays")


class CountersStruct(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]


class TimingStruct(C.Structure):
    _fields_ = [("ms", C.c_double * len(STAGES)), ("launches", C.c_uint64 * len(STAGES))]


_lib = None


def lib():
    """Loads librthip.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RtHipError(-1, f"{_LIB_PATH} is missing: run __graft_entry__.build() first")
        l = C.CDLL(_LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        l.rthip_device_count.restype = C.c_int
        l.rthip_create.argtypes = [C.c_int, C.POINTER(vp)]
        l.rthip_destroy.argtypes = [vp]
        l.rthip_destroy.restype = None
        l.rthip_last_error.argtypes = [vp]
        l.rthip_last_error.restype = C.c_char_p
        l.rthip_set_stream.argtypes = [vp, vp]
        l.rthip_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
        l.rthip_set_frame_uniforms.argtypes = [vp, vp]
        l.rthip_default_params.argtypes = [C.POINTER(RenderParams)]
        l.rthip_reserve.argtypes = [vp, C.c_uint32]
        l.rthip_frames_in_flight.argtypes = [vp, C.POINTER(C.c_uint32)]
        l.rthip_candidate_walk.argtypes = [vp, C.POINTER(C.c_uint32)]
        l.rthip_trace_rays.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(RenderParams), vp, vp, vp, vp]
        l.rthip_trace_rays_noalloc.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(RenderParams), vp, vp, vp, vp]
        l.rthip_copy_bandwidth.argtypes = [vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
        l.rthip_gather_ceiling.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        l.rthip_default_params.restype = None
        l.rthip_render.argtypes = [vp, u32, u32, u32, C.POINTER(RenderParams)]
        l.rthip_bind_accum.argtypes = [vp, vp, u64]
        l.rthip_accum_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        l.rthip_read_accum.argtypes = [vp, vp, u64]
        l.rthip_write_accum.argtypes = [vp, vp, u64]
        l.rthip_read_taps.argtypes = [vp, vp, vp, vp, vp, u64]
        l.rthip_read_nodes_device_layout.argtypes = [vp, vp, u64, C.POINTER(u64)]
        l.rthip_read_node_index_map.argtypes = [vp, vp, u64]
        l.rthip_read_tri_order.argtypes = [vp, vp, u64]
        l.rthip_read_tris_device_layout.argtypes = [vp, vp, u64]
        l.rthip_resolve.argtypes = [vp, vp, u64, C.c_int]
        l.rthip_get_counters.argtypes = [vp, C.POINTER(CountersStruct)]
        l.rthip_reset_counters.argtypes = [vp]
        l.rthip_get_timing.argtypes = [vp, C.POINTER(TimingStruct)]
        l.rthip_reset_timing.argtypes = [vp]
        l.rthip_synchronize.argtypes = [vp]
        _lib = l
    return _lib


def device_count():
    return int(lib().rthip_device_count())


def default_params(**kw):
    p = RenderParams()
    lib().rthip_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU.  Not thread-safe (one context per thread / process)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib().rthip_create(device, C.byref(h))
        if rc != 0:
            raise RtHipError(rc, lib().rthip_last_error(None).decode())
        self._h = h
        self.res = None
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            lib().rthip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RtHipError(rc, lib().rthip_last_error(self._h).decode())

    # -- main.rs:328-353 ----------------------------------------------------------------
    def upload_scene(self, scene):
        """scene: a raytracer_amd.host.Scene or a dict of arrays (tris, tri_exts, nodes, prims,
        texels, env[h,w,4])."""
        a = scene if isinstance(scene, dict) else scene.arrays()
        tris = np.ascontiguousarray(a["tris"])
        exts = np.ascontiguousarray(a["tri_exts"])
        nodes = np.ascontiguousarray(a["nodes"])
        prims = np.ascontiguousarray(a["prims"])
        texels = np.ascontiguousarray(a["texels"], dtype=np.uint32)
        env = np.ascontiguousarray(a["env"], dtype=np.float32)
        assert tris.nbytes % 48 == 0 and exts.nbytes == tris.nbytes * 2
        assert nodes.nbytes % 32 == 0 and prims.nbytes % 240 == 0
        assert env.ndim == 3 and env.shape[2] == 4
        d = SceneDesc()
        d.tris, d.n_tris, d.tri_exts = _p(tris), tris.nbytes // 48, _p(exts)
        d.nodes, d.n_nodes = _p(nodes), nodes.nbytes // 32
        d.prims, d.n_prims = _p(prims), prims.nbytes // 240
        d.texels, d.n_texels = _p(texels), texels.size
        d.env, d.env_w, d.env_h = _p(env), env.shape[1], env.shape[0]
        self._check(lib().rthip_upload_scene(self._h, C.byref(d)))
        self.n_tris, self.n_nodes = int(d.n_tris), int(d.n_nodes)

    # -- main.rs:372-380 ----------------------------------------------------------------
    def set_frame_uniforms(self, uniforms):
        u = np.ascontiguousarray(uniforms).view(np.uint8).reshape(-1)
        assert u.size == 608
        self._check(lib().rthip_set_frame_uniforms(self._h, _p(u)))
        w, h = (int(v) for v in u[576:584].view(np.uint32))
        self.res = (w, h)

    # -- main.rs:383-393 ----------------------------------------------------------------
    def render(self, first_frame=1, n_frames=1, frame_stride=1, params=None):
        p = params or default_params()
        self._check(lib().rthip_render(self._h, first_frame, n_frames, frame_stride, C.byref(p)))

    def trace_rays(self, origins, directions, seeds=None, params=None):
        """trace() (shader.wgsl:598-627) for arbitrary rays.  Returns dict(t, tri, prim, overflow, seed)."""
        o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        n = o.shape[0]
        rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
        sd = np.zeros(n, np.uint32) if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        prim = np.zeros(n, np.int32)
        ovf = np.zeros(n, np.uint8)
        p = params or default_params()
        self._check(lib().rthip_trace_rays(self._h, _p(rays), _p(sd), n, C.byref(p), _p(t), _p(tri), _p(prim), _p(ovf)))
        return dict(t=t, tri=tri, prim=prim, overflow=ovf, seed=sd)

    def trace_rays_noalloc(self, d_rays6, d_seeds, n, d_t, d_tri, d_prim, d_overflow, params=None):
        """rthip_trace_rays_noalloc: every argument but n and params is a DEVICE address (int, 0 =
        NULL) of caller-owned memory on this context's GPU; asynchronous on the context's stream;
        at most NOALLOC_MAX_RAYS rays."""
        p = params or default_params()
        v = lambda x: C.c_void_p(x) if x else None  # noqa: E731
        self._check(lib().rthip_trace_rays_noalloc(self._h, v(d_rays6), v(d_seeds), int(n), C.byref(p),
                                                   v(d_t), v(d_tri), v(d_prim), v(d_overflow)))

    def reserve(self, n_frames):
        """Allocate the work buffers for calls of up to n_frames frames ahead of the first render."""
        self._check(lib().rthip_reserve(self._h, int(n_frames)))

    def frames_in_flight(self):
        """Frames the next render moves through the stages together (after the memory budget)."""
        n = C.c_uint32()
        self._check(lib().rthip_frames_in_flight(self._h, C.byref(n)))
        return int(n.value)

    def candidate_walk(self):
        """How the uploaded scene's primitives are enumerated per ray: "mask", "scan", "pyramid" or "list"."""
        k = C.c_uint32()
        self._check(lib().rthip_candidate_walk(self._h, C.byref(k)))
        return ("mask", "scan", "pyramid", "list")[k.value]

    def copy_bandwidth(self, nbytes=1 << 30, repeats=8):
        """Device-to-device stream-copy ceiling in GB/s (read + write traffic)."""
        out = C.c_double()
        self._check(lib().rthip_copy_bandwidth(self._h, int(nbytes), int(repeats), C.byref(out)))
        return float(out.value)

    def gather_ceiling(self, active_lanes, record_bytes=64, records_in_flight=1):
        """Lane-loads (16 bytes each) per second that the chip's vector-memory pipelines serve out of
        the L1 for `active_lanes` scattered lanes per wave reading one or two records of 64 bytes (the
        inner-node step's shape) or 128 bytes each per step."""
        out = C.c_double()
        self._check(lib().rthip_gather_ceiling(self._h, int(active_lanes), int(record_bytes), int(records_in_flight), C.byref(out)))
        return float(out.value)

    def synchronize(self):
        self._check(lib().rthip_synchronize(self._h))

    def set_stream(self, stream_handle):
        self._check(lib().rthip_set_stream(self._h, C.c_void_p(stream_handle) if stream_handle else None))

    def bind_accum(self, device_ptr, n_pixels):
        self._check(lib().rthip_bind_accum(self._h, C.c_void_p(device_ptr) if device_ptr else None, n_pixels))

    def accum_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(lib().rthip_accum_device_ptr(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def read_accum(self):
        w, h = self.res
        out = np.zeros((h, w, 4), dtype=np.float32)
        self._check(lib().rthip_read_accum(self._h, _p(out), w * h))
        return out

    def write_accum(self, rgba):
        w, h = self.res
        a = np.ascontiguousarray(rgba, dtype=np.float32).reshape(h, w, 4)
        self._check(lib().rthip_write_accum(self._h, _p(a), w * h))

    def read_taps(self):
        w, h = self.res
        tri = np.zeros((h, w), dtype=np.int32)
        prim = np.zeros((h, w), dtype=np.int32)
        t = np.zeros((h, w), dtype=np.float32)
        seed = np.zeros((h, w), dtype=np.uint32)
        self._check(lib().rthip_read_taps(self._h, _p(tri), _p(prim), _p(t), _p(seed), w * h))
        return dict(tri=tri, prim=prim, t=t, seed=seed)

    def read_nodes_device_layout(self):
        n = C.c_uint64()
        self._check(lib().rthip_read_nodes_device_layout(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=L.NODE)
        self._check(lib().rthip_read_nodes_device_layout(self._h, _p(out), n.value, C.byref(n)))
        return out

    def read_node_index_map(self):
        out = np.zeros(self.n_nodes, dtype=np.uint32)
        self._check(lib().rthip_read_node_index_map(self._h, _p(out), self.n_nodes))
        return out

    def read_tri_order(self):
        out = np.zeros(self.n_tris, dtype=np.uint32)
        self._check(lib().rthip_read_tri_order(self._h, _p(out), self.n_tris))
        return out

    def read_tris_device_layout(self):
        out = np.zeros((self.n_tris, 3, 4), dtype=np.float32)
        self._check(lib().rthip_read_tris_device_layout(self._h, _p(out), self.n_tris))
        return out

    # -- main.rs:395-401 ----------------------------------------------------------------
    def resolve(self, apply_srgb_oetf=True):
        w, h = self.res
        out = np.zeros((h, w, 4), dtype=np.uint8)
        self._check(lib().rthip_resolve(self._h, _p(out), w * h, 1 if apply_srgb_oetf else 0))
        return out

    def resolve_into(self, ptr, apply_srgb_oetf=True):
        """The image into memory the caller names by address: host memory, or memory of this GPU
        (w * h * 4 bytes, 4-byte aligned), which the kernel then writes in place."""
        w, h = self.res
        self._check(lib().rthip_resolve(self._h, C.c_void_p(int(ptr)), w * h, 1 if apply_srgb_oetf else 0))

    def counters(self):
        c = CountersStruct()
        self._check(lib().rthip_get_counters(self._h, C.byref(c)))
        return {n: int(getattr(c, n)) for n in COUNTER_NAMES}

    def reset_counters(self):
        self._check(lib().rthip_reset_counters(self._h))

    def timing(self):
        t = TimingStruct()
        self._check(lib().rthip_get_timing(self._h, C.byref(t)))
        return {s: dict(ms=float(t.ms[i]), launches=int(t.launches[i])) for i, s in enumerate(STAGES)}

    def reset_timing(self):
        self._check(lib().rthip_reset_timing(self._h))
