"""Python face of include/rthip_group.h (librthip_group.so): several GPUs of one node driven
from one process, frames striped over them, one RCCL reduce of the accumulators.

The samples of the reference's frame loop (main.rs:356-407) are independent given
(pixel, frame) (shader.wgsl:895); GPU r of n renders frames first + r, first + r + n, ...
No fallback: without the library every call raises.
"""
import ctypes as C
import os

import numpy as np

from . import device

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RTHIP_GROUP_LIBRARY") or os.path.join(_HERE, "lib", "librthip_group.so")

EXPORTS = (
    "rthip_group_create", "rthip_group_destroy", "rthip_group_last_error", "rthip_group_size",
    "rthip_group_ctx", "rthip_group_upload_scene", "rthip_group_set_frame_uniforms",
    "rthip_group_plan", "rthip_group_render", "rthip_group_reduce", "rthip_group_synchronize",
)

_lib = None


def lib():
    global _lib
    if _lib is None:
        device.lib()  # the single-GPU library first, so that both share one copy of it
        if not os.path.exists(_LIB_PATH):
            raise device.RtHipError(-1, f"{_LIB_PATH} is missing: run __graft_entry__.build() first")
        l = C.CDLL(_LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        l.rthip_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        l.rthip_group_destroy.argtypes = [vp]
        l.rthip_group_destroy.restype = None
        l.rthip_group_last_error.argtypes = [vp]
        l.rthip_group_last_error.restype = C.c_char_p
        l.rthip_group_size.argtypes = [vp]
        l.rthip_group_ctx.argtypes = [vp, C.c_int]
        l.rthip_group_ctx.restype = vp
        l.rthip_group_upload_scene.argtypes = [vp, C.POINTER(device.SceneDesc)]
        l.rthip_group_set_frame_uniforms.argtypes = [vp, vp]
        l.rthip_group_plan.argtypes = [u32, u32, C.c_int, C.c_int, C.POINTER(u32), C.POINTER(u32)]
        l.rthip_group_plan.restype = None
        l.rthip_group_render.argtypes = [vp, u32, u32, C.POINTER(device.RenderParams)]
        l.rthip_group_reduce.argtypes = [vp, C.c_int]
        l.rthip_group_synchronize.argtypes = [vp]
        _lib = l
    return _lib


def plan(first_frame, n_frames, n_ranks, rank):
    """(first, count): rank renders frames first, first + n_ranks, ... (count of them)."""
    f, c = C.c_uint32(), C.c_uint32()
    lib().rthip_group_plan(first_frame, n_frames, n_ranks, rank, C.byref(f), C.byref(c))
    return int(f.value), int(c.value)


class _BorrowedContext(device.Context):
    """A context owned by a group: same methods, never destroyed from here."""

    def __init__(self, handle):  # noqa: super().__init__ would create a context
        self._h = C.c_void_p(handle)
        self.res = None
        self._keep = None

    def close(self):
        self._h = None


class Group:
    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = lib().rthip_group_create(devs, len(devices), C.byref(h))
        if rc != 0:
            raise device.RtHipError(rc, lib().rthip_group_last_error(None).decode())
        self._h = h
        self.size = int(lib().rthip_group_size(h))
        self.contexts = [_BorrowedContext(lib().rthip_group_ctx(h, r)) for r in range(self.size)]

    def close(self):
        if getattr(self, "_h", None):
            for c in self.contexts:
                c.close()
            lib().rthip_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise device.RtHipError(rc, lib().rthip_group_last_error(self._h).decode())

    def upload_scene(self, scene):
        a = scene if isinstance(scene, dict) else scene.arrays()
        tris = np.ascontiguousarray(a["tris"])
        exts = np.ascontiguousarray(a["tri_exts"])
        nodes = np.ascontiguousarray(a["nodes"])
        prims = np.ascontiguousarray(a["prims"])
        texels = np.ascontiguousarray(a["texels"], dtype=np.uint32)
        env = np.ascontiguousarray(a["env"], dtype=np.float32)
        d = device.SceneDesc()
        d.tris, d.n_tris, d.tri_exts = device._p(tris), tris.nbytes // 48, device._p(exts)
        d.nodes, d.n_nodes = device._p(nodes), nodes.nbytes // 32
        d.prims, d.n_prims = device._p(prims), prims.nbytes // 240
        d.texels, d.n_texels = device._p(texels), texels.size
        d.env, d.env_w, d.env_h = device._p(env), env.shape[1], env.shape[0]
        self._check(lib().rthip_group_upload_scene(self._h, C.byref(d)))
        for c in self.contexts:
            c.n_tris, c.n_nodes = int(d.n_tris), int(d.n_nodes)

    def set_frame_uniforms(self, uniforms):
        u = np.ascontiguousarray(uniforms).view(np.uint8).reshape(-1)
        assert u.size == 608
        self._check(lib().rthip_group_set_frame_uniforms(self._h, device._p(u)))
        w, h = (int(v) for v in u[576:584].view(np.uint32))
        for c in self.contexts:
            c.res = (w, h)

    def render(self, first_frame=1, n_frames=1, params=None):
        p = params or device.default_params()
        self._check(lib().rthip_group_render(self._h, first_frame, n_frames, C.byref(p)))

    def reduce(self, root=0):
        self._check(lib().rthip_group_reduce(self._h, root))

    def synchronize(self):
        self._check(lib().rthip_group_synchronize(self._h))
