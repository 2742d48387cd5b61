"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing in the
shipped package may import it.  Parity unpinned: see oracle/rt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librt_oracle.so")


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("rt_oracle.c", "rt_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class Scene(C.Structure):
    _fields_ = [
        ("tris", C.c_void_p), ("n_tris", C.c_uint64),
        ("exts", C.c_void_p),
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
        ("prims", C.c_void_p), ("n_prims", C.c_uint32),
        ("texels", C.c_void_p), ("n_texels", C.c_uint64),
        ("env", C.c_void_p), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("max_bounces", C.c_uint32), ("term_prob", C.c_float), ("aperture_radius", C.c_float),
        ("stack_depth", C.c_uint32), ("enable_sun", C.c_uint32),
    ]


COUNTER_NAMES = (
    "rays", "root_tests", "inner_steps", "leaf_steps", "tri_tests", "alpha_tests",
    "texel_fetches", "shade_calls", "misses", "kills", "rng_draws", "stack_overflows",
    "shadow_rays",
)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_hash21.restype = C.c_uint32
        L.orc_hash21.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_rand_u32.restype = C.c_uint32
        L.orc_rand_u32.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_rand.restype = C.c_float
        L.orc_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_pixel_seed.restype = C.c_uint32
        L.orc_pixel_seed.argtypes = [C.c_uint32] * 3
        L.orc_bvh_build.restype = C.c_uint32
        L.orc_bvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_build_primitive.restype = C.c_uint32
        L.orc_build_primitive.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_tri_new.argtypes = [C.c_void_p] * 4
        L.orc_pack_oct.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pack_alpha.restype = C.c_uint32
        L.orc_pack_alpha.argtypes = [C.c_uint32, C.c_float]
        L.orc_mat4_inverse_f64.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_camera_default.argtypes = [C.c_void_p]
        L.orc_camera_dir.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.orc_render.argtypes = [C.POINTER(Scene), C.c_void_p, C.POINTER(Params), C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.orc_render_rows.argtypes = L.orc_render.argtypes + [C.c_uint32, C.c_uint32]
        L.orc_trace_ray.restype = C.c_int32
        L.orc_trace_ray.argtypes = [C.POINTER(Scene), C.POINTER(Params), C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                    C.POINTER(C.c_int32)]
        L.orc_tonemap_pbr_neutral.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_resolve.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def hash21(x, y):
    return int(lib().orc_hash21(x & 0xFFFFFFFF, y & 0xFFFFFFFF))


def pixel_seed(px, py, frame):
    return int(lib().orc_pixel_seed(px, py, frame))


def rand_stream(seed, n):
    """Returns ([(seed_after, out_u32)] * n) for the rand() recurrence."""
    s = C.c_uint32(seed)
    out = []
    for _ in range(n):
        r = lib().orc_rand_u32(C.byref(s))
        out.append((int(s.value), int(r)))
    return out


def rand_float(seed):
    s = C.c_uint32(seed)
    return float(lib().orc_rand(C.byref(s))), int(s.value)


def default_params(**kw):
    p = Params(8, 0.5, 0.025, 23, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def bvh_build(tris, offset, size):
    """tris: contiguous array of 48-byte triangle records (any dtype).  Returns
    (nodes as float32[n,8] view-compatible bytes, indices uint32[size])."""
    assert tris.flags.c_contiguous and tris.nbytes % 48 == 0
    nodes = np.zeros((max(2 * size, 1), 8), dtype=np.float32)
    indices = np.zeros(max(size, 1), dtype=np.uint32)
    n = lib().orc_bvh_build(_ptr(tris), offset, size, _ptr(nodes), _ptr(indices))
    return nodes[:n].copy(), indices[:size].copy()


def build_primitive(tris, exts, offset, size, bvh_root):
    """In-place flatten of tris/exts[offset:offset+size]; returns (nodes[n,8] f32, order)."""
    assert tris.flags.c_contiguous and exts.flags.c_contiguous
    nodes = np.zeros((max(2 * size, 1), 8), dtype=np.float32)
    order = np.zeros(max(size, 1), dtype=np.uint32)
    n = lib().orc_build_primitive(_ptr(tris), _ptr(exts), offset, size, _ptr(nodes), bvh_root,
                                  _ptr(order))
    return nodes[:n].copy(), order[:size].copy()


def make_scene(tris, exts, nodes, prims, texels, env, env_w, env_h):
    """All arguments are contiguous numpy arrays holding the reference's byte layouts."""
    keep = [np.ascontiguousarray(a) if a is not None else None
            for a in (tris, exts, nodes, prims, texels, env)]
    tris, exts, nodes, prims, texels, env = keep
    s = Scene()
    s.tris, s.n_tris = _ptr(tris), tris.nbytes // 48
    s.exts = _ptr(exts)
    s.nodes, s.n_nodes = _ptr(nodes), nodes.nbytes // 32
    s.prims, s.n_prims = _ptr(prims), prims.nbytes // 240
    s.texels = _ptr(texels)
    s.n_texels = 0 if texels is None else texels.size
    s.env, s.env_w, s.env_h = _ptr(env), env_w, env_h
    s._keep = keep
    return s


def render(scene, uniforms, params=None, first_frame=1, n_frames=1, frame_stride=1,
           taps=False, n_threads=1, accum=None, row_start=0, row_step=1):
    """uniforms: 608-byte numpy buffer.  Returns dict(accum, counters[, tri, prim, t, seed])."""
    params = params or default_params()
    u = np.ascontiguousarray(uniforms).view(np.uint8).reshape(-1)
    assert u.size == 608
    w, h = (int(v) for v in u[576:584].view(np.uint32))
    if accum is None:
        accum = np.zeros((h, w, 4), dtype=np.float32)
    out = {"accum": accum}
    tri = prim = t = seed = None
    if taps:
        tri = np.full((h, w), -2, dtype=np.int32)
        prim = np.full((h, w), -2, dtype=np.int32)
        t = np.zeros((h, w), dtype=np.float32)
        seed = np.zeros((h, w), dtype=np.uint32)
        out.update(tri=tri, prim=prim, t=t, seed=seed)
    c = Counters()
    lib().orc_render_rows(C.byref(scene), _ptr(u), C.byref(params), first_frame, n_frames,
                          frame_stride, _ptr(accum), _ptr(tri), _ptr(prim), _ptr(t), _ptr(seed),
                          C.byref(c), n_threads, row_start, row_step)
    out["counters"] = c.as_dict()
    return out


def trace_ray(scene, origin, direction, params=None, seed=0):
    params = params or default_params()
    o = np.asarray(origin, dtype=np.float32)
    d = np.asarray(direction, dtype=np.float32)
    s = C.c_uint32(seed)
    t = C.c_float(0)
    p = C.c_int32(-1)
    tri = lib().orc_trace_ray(C.byref(scene), C.byref(params), _ptr(o), _ptr(d), C.byref(s),
                              C.byref(t), C.byref(p))
    return int(tri), int(p.value), float(t.value), int(s.value)


def resolve(accum, apply_srgb_oetf=True):
    a = np.ascontiguousarray(accum, dtype=np.float32)
    n = a.size // 4
    out = np.zeros(a.shape[:-1] + (4,), dtype=np.uint8)
    lib().orc_resolve(_ptr(a), n, _ptr(out), 1 if apply_srgb_oetf else 0)
    return out


def tonemap(rgb):
    i = np.asarray(rgb, dtype=np.float32).copy()
    o = np.zeros(3, dtype=np.float32)
    lib().orc_tonemap_pbr_neutral(_ptr(i), _ptr(o))
    return o


def pack_oct(n):
    i = np.asarray(n, dtype=np.float32).copy()
    o = np.zeros(2, dtype=np.float32)
    lib().orc_pack_oct(_ptr(i), _ptr(o))
    return o


def mat4_inverse_f64(m):
    i = np.ascontiguousarray(m, dtype=np.float32).reshape(16).copy()
    o = np.zeros(16, dtype=np.float32)
    lib().orc_mat4_inverse_f64(_ptr(i), _ptr(o))
    return o


def camera_default():
    o = np.zeros(12, dtype=np.float32)
    lib().orc_camera_default(_ptr(o))
    return o
