"""Python face of the host-side scene pipeline (include/rthost.h, librthost.so).

Mirrors the reference's `Scene` (scene.rs:223-710) and `Camera` (input.rs:34-254): the same
operations under the same names, producing the flat arrays the device library consumes.
"""
import ctypes as C
import os

import numpy as np

from . import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "librthost.so")


class RtHostError(RuntimeError):
    pass


class _View(C.Structure):
    _fields_ = [
        ("tris", C.c_void_p), ("n_tris", C.c_uint64),
        ("tri_exts", C.c_void_p),
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
        ("prims", C.c_void_p), ("n_prims", C.c_uint32),
        ("texels", C.c_void_p), ("n_texels", C.c_uint64),
        ("env", C.c_void_p), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
        ("tri_order", C.c_void_p),
        ("n_cameras", C.c_uint32),
    ]


class CameraStruct(C.Structure):
    _fields_ = [
        ("yaw", C.c_float), ("pitch", C.c_float), ("roll", C.c_float),
        ("position", C.c_float * 3),
        ("fovy", C.c_float), ("focus", C.c_float),
        ("aperture", C.c_float), ("exposure", C.c_float), ("bloom", C.c_float),
        ("dispersion", C.c_float),
        ("aspect", C.c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RtHostError(f"{_LIB_PATH} is missing: run __graft_entry__.build() first")
        l = C.CDLL(_LIB_PATH)
        l.rthost_last_error.restype = C.c_char_p
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        l.rthost_scene_new.argtypes = [C.POINTER(vp)]
        l.rthost_scene_free.argtypes = [vp]
        l.rthost_scene_free.restype = None
        l.rthost_scene_view_get.argtypes = [vp, C.POINTER(_View)]
        l.rthost_scene_add_texture.argtypes = [vp, vp, u32, u32, vp]
        l.rthost_scene_add_primitive.argtypes = [vp, vp, u32, vp, u64, vp, vp, vp, vp, vp, vp,
                                                 C.POINTER(u32)]
        l.rthost_scene_add_instance.argtypes = [vp, u32, vp, C.POINTER(u32)]
        l.rthost_scene_set_env.argtypes = [vp, vp, u32, u32]
        l.rthost_scene_load_gltf.argtypes = [vp, vp, C.c_size_t, C.c_char_p]
        l.rthost_scene_load_gltf_file.argtypes = [vp, C.c_char_p]
        l.rthost_scene_load_hdr_file.argtypes = [vp, C.c_char_p]
        l.rthost_scene_save.argtypes = [vp, C.c_char_p]
        l.rthost_scene_load_cache.argtypes = [vp, C.c_char_p]
        l.rthost_camera_default.argtypes = [C.POINTER(CameraStruct)]
        l.rthost_camera_default.restype = None
        l.rthost_camera_to_gpu.argtypes = [C.POINTER(CameraStruct), vp]
        l.rthost_camera_to_gpu.restype = None
        l.rthost_scene_add_camera.argtypes = [vp, C.POINTER(CameraStruct)]
        l.rthost_scene_get_camera.argtypes = [vp, u32, C.POINTER(CameraStruct)]
        l.rthost_scene_focus_camera.argtypes = [vp, u32, C.POINTER(C.c_int)]
        l.rthost_frame_uniforms.argtypes = [vp, u32, vp, u32, u32, u32, C.c_float, u32, vp]
        l.rthost_bvh_build.argtypes = [vp, u32, u32, vp, vp, C.POINTER(u32), C.c_int]
        l.rthost_tri_new.argtypes = [vp, vp, vp, vp]
        l.rthost_tri_new.restype = None
        l.rthost_pack_oct.argtypes = [vp, vp]
        l.rthost_pack_oct.restype = None
        l.rthost_pack_alpha.argtypes = [u32, C.c_float]
        l.rthost_pack_alpha.restype = u32
        for n in ("rthost_from_gltf_mat4", "rthost_mat4_inverse_f64", "rthost_mat4_inverse_f32"):
            getattr(l, n).argtypes = [vp, vp]
            getattr(l, n).restype = None
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise RtHostError(f"rthost error {rc}: {lib().rthost_last_error().decode()}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


def translation(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return np.ascontiguousarray(m.T).reshape(16)  # column-major


def trs(t=(0, 0, 0), rot_z=0.0, scale=(1, 1, 1), rot_y=0.0, rot_x=0.0):
    """Column-major float[16] of T * Rz * Ry * Rx * S (world axes)."""
    cz, sz = np.cos(rot_z), np.sin(rot_z)
    cy, sy = np.cos(rot_y), np.sin(rot_y)
    cx, sx = np.cos(rot_x), np.sin(rot_x)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    m = np.eye(4)
    m[:3, :3] = rz @ ry @ rx @ np.diag(scale)
    m[:3, 3] = t
    return np.ascontiguousarray(m.T, dtype=np.float32).reshape(16)


def basis(t, cx, cy, cz):
    """Column-major float[16] with the given images of the local x, y, z axes and translation t."""
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = cx, cy, cz, t
    return np.ascontiguousarray(m.T, dtype=np.float32).reshape(16)


class Camera:
    """input.rs:34-51.  Angles in radians; world is X forward, Z up."""

    def __init__(self, struct=None):
        if struct is None:
            struct = CameraStruct()
            lib().rthost_camera_default(C.byref(struct))
        self.c = struct

    @classmethod
    def default(cls):  # input.rs:202-219
        return cls()

    @classmethod
    def look(cls, position, yaw=0.0, pitch=0.0, fovy=np.radians(90.0), focus=1.0):
        cam = cls()
        cam.c.position[:] = [float(v) for v in position]
        cam.c.yaw, cam.c.pitch, cam.c.roll = float(yaw), float(pitch), 0.0
        cam.c.fovy, cam.c.focus = float(fovy), float(focus)
        return cam

    def to_gpu(self):  # input.rs:100-111
        out = np.zeros((), dtype=L.GPU_CAMERA)
        lib().rthost_camera_to_gpu(C.byref(self.c), _p(out))
        return out


class Scene:
    """scene.rs:223-252.  Array properties return copies in the reference's byte layouts."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib().rthost_scene_new(C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rthost_scene_free(self._h)
            self._h = None

    # -- construction -------------------------------------------------------------------
    @classmethod
    def from_path(cls, mesh_path, env_map_path=None):  # scene.rs:680-708
        s = cls()
        _check(lib().rthost_scene_load_gltf_file(s._h, os.fsencode(mesh_path)))
        if env_map_path:
            _check(lib().rthost_scene_load_hdr_file(s._h, os.fsencode(env_map_path)))
        if s.n_cameras == 0:
            s.add_camera(Camera.default())
        return s

    @classmethod
    def from_cache(cls, path):
        """A scene written by save() (or by the reference's host in the same layout)."""
        s = cls()
        _check(lib().rthost_scene_load_cache(s._h, os.fsencode(path)))
        return s

    def save(self, path):
        """Write the flat arrays and cameras as one .rtscene file."""
        _check(lib().rthost_scene_save(self._h, os.fsencode(path)))

    def add_gltf_bytes(self, data, base_dir="."):  # scene.rs:268-284
        buf = np.frombuffer(data, dtype=np.uint8)
        _check(lib().rthost_scene_load_gltf(self._h, _p(buf), buf.size, os.fsencode(base_dir)))

    def add_texture(self, rgba8):
        """rgba8: (h, w, 4) uint8.  Returns a TEX_REF record."""
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 4
        ref = np.zeros(2, dtype=np.uint32)
        _check(lib().rthost_scene_add_texture(self._h, _p(img), img.shape[1], img.shape[0], _p(ref)))
        r = np.zeros((), dtype=L.TEX_REF)
        r["offset"], r["size"] = ref
        return r

    def add_primitive(self, positions, indices, normals=None, uv0=None, uv1=None, colors=None,
                      material=None, transform=IDENTITY):
        pos = _f32(positions, (-1, 3))
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        nrm = _f32(normals, (-1, 3))
        t0, t1 = _f32(uv0, (-1, 2)), _f32(uv1, (-1, 2))
        col = None if colors is None else np.ascontiguousarray(colors, dtype=np.uint32)
        for a in (nrm, t0, t1, col):
            assert a is None or a.shape[0] == pos.shape[0]
        mat = np.ascontiguousarray(L.default_material() if material is None else material)
        assert mat.nbytes == 96
        xf = _f32(transform, 16)
        out = C.c_uint32()
        _check(lib().rthost_scene_add_primitive(self._h, _p(pos), pos.shape[0], _p(idx), idx.size,
                                                _p(nrm), _p(t0), _p(t1), _p(col), _p(mat), _p(xf),
                                                C.byref(out)))
        return int(out.value)

    def add_instance(self, prim, transform):
        out = C.c_uint32()
        _check(lib().rthost_scene_add_instance(self._h, prim, _p(_f32(transform, 16)), C.byref(out)))
        return int(out.value)

    def set_env(self, rgba32f):
        env = _f32(rgba32f)
        assert env.ndim == 3 and env.shape[2] == 4
        _check(lib().rthost_scene_set_env(self._h, _p(env), env.shape[1], env.shape[0]))

    def set_equirectangular_env_map(self, path):  # scene.rs:603-615
        _check(lib().rthost_scene_load_hdr_file(self._h, os.fsencode(path)))

    def add_camera(self, cam):
        _check(lib().rthost_scene_add_camera(self._h, C.byref(cam.c)))

    def camera(self, idx=0):
        c = CameraStruct()
        _check(lib().rthost_scene_get_camera(self._h, idx, C.byref(c)))
        return Camera(c)

    def focus_camera(self, idx=0):  # scene.rs:670-677
        hit = C.c_int()
        _check(lib().rthost_scene_focus_camera(self._h, idx, C.byref(hit)))
        return bool(hit.value)

    # -- flat arrays --------------------------------------------------------------------
    def _view(self):
        v = _View()
        _check(lib().rthost_scene_view_get(self._h, C.byref(v)))
        return v

    @staticmethod
    def _copy(ptr, count, dtype):
        if not ptr or count == 0:
            return np.zeros(0, dtype=dtype)
        n = count * np.dtype(dtype).itemsize
        # one copy, and no 2 GiB limit (C.string_at goes through a Python bytes object)
        return np.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype).copy()

    @property
    def n_cameras(self):
        return int(self._view().n_cameras)

    @property
    def tris(self):
        v = self._view()
        return self._copy(v.tris, v.n_tris, L.TRI)

    @property
    def tri_exts(self):
        v = self._view()
        return self._copy(v.tri_exts, v.n_tris, L.TRI_EXT)

    @property
    def bvh_node_data(self):
        v = self._view()
        return self._copy(v.nodes, v.n_nodes, L.NODE)

    @property
    def primitives(self):
        v = self._view()
        return self._copy(v.prims, v.n_prims, L.PRIMITIVE)

    @property
    def texture_data(self):
        v = self._view()
        return self._copy(v.texels, v.n_texels, np.uint32)

    @property
    def env_map_data(self):
        v = self._view()
        return self._copy(v.env, v.env_w * v.env_h * 4, np.float32).reshape(v.env_h, v.env_w, 4)

    @property
    def tri_order(self):
        v = self._view()
        return self._copy(v.tri_order, v.n_tris, np.uint32)

    def arrays(self):
        """One consistent snapshot of every array (dict)."""
        v = self._view()
        return dict(
            tris=self._copy(v.tris, v.n_tris, L.TRI),
            tri_exts=self._copy(v.tri_exts, v.n_tris, L.TRI_EXT),
            nodes=self._copy(v.nodes, v.n_nodes, L.NODE),
            prims=self._copy(v.prims, v.n_prims, L.PRIMITIVE),
            texels=self._copy(v.texels, v.n_texels, np.uint32),
            env=self._copy(v.env, v.env_w * v.env_h * 4, np.float32).reshape(v.env_h, v.env_w, 4),
            tri_order=self._copy(v.tri_order, v.n_tris, np.uint32),
        )

    def frame_uniforms(self, res, frame=1, reject_hist=1, camera_idx=0, gpu_camera=None, time=0.0):
        """FrameUniforms as frame() writes it (main.rs:372-380).  Returns a FRAME_UNIFORMS record."""
        out = np.zeros((), dtype=L.FRAME_UNIFORMS)
        cam = None if gpu_camera is None else np.ascontiguousarray(gpu_camera)
        _check(lib().rthost_frame_uniforms(self._h, camera_idx, _p(cam), int(res[0]), int(res[1]),
                                           int(frame), float(time), int(reject_hist), _p(out)))
        return out


def bvh_build(tris, offset=0, size=None, n_threads=0):
    """Bvh::new over tris[offset:offset+size] (scene.rs:888-899).  Returns (nodes, indices)."""
    tris = np.ascontiguousarray(tris)
    assert tris.nbytes % 48 == 0
    if size is None:
        size = tris.nbytes // 48 - offset
    nodes = np.zeros(max(2 * size, 1), dtype=L.NODE)
    idx = np.zeros(max(size, 1), dtype=np.uint32)
    n = C.c_uint32()
    _check(lib().rthost_bvh_build(_p(tris), offset, size, _p(nodes), _p(idx), C.byref(n), n_threads))
    return nodes[: n.value].copy(), idx[:size].copy()


def tri_new(p1, p2, p3):
    out = np.zeros((), dtype=L.TRI)
    a, b, c = (_f32(p, 3) for p in (p1, p2, p3))
    lib().rthost_tri_new(_p(a), _p(b), _p(c), _p(out))
    return out


def make_tris(positions, indices):
    """Vectorised Tri::new (scene.rs:720-725): centroid = ((p1 + p2) + p3) / 3 in f32."""
    pos = _f32(positions, (-1, 3))
    idx = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    p = pos[idx]  # (n, 3 verts, 3)
    out = np.zeros(idx.shape[0], dtype=L.TRI)
    out["v"][:, :, :3] = p
    c = ((p[:, 0] + p[:, 1]) + p[:, 2]) / np.float32(3.0)
    out["v"][:, :, 3] = c
    return out


def pack_oct(n):
    o = np.zeros(2, dtype=np.float32)
    lib().rthost_pack_oct(_p(_f32(n, 3)), _p(o))
    return o


def pack_alpha(mode, cutoff):
    return int(lib().rthost_pack_alpha(mode, cutoff))


def mat4_inverse_f64(m):
    o = np.zeros(16, dtype=np.float32)
    lib().rthost_mat4_inverse_f64(_p(_f32(m, 16)), _p(o))
    return o


def mat4_inverse_f32(m):
    o = np.zeros(16, dtype=np.float32)
    lib().rthost_mat4_inverse_f32(_p(_f32(m, 16)), _p(o))
    return o


def from_gltf_mat4(m):
    o = np.zeros(16, dtype=np.float32)
    lib().rthost_from_gltf_mat4(_p(_f32(m, 16)), _p(o))
    return o
