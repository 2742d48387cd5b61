"""Tiny independent glTF/GLB reader used only by tests to cross-check the C++ loader.

Reads POSITION / indices / node matrices of every triangle primitive, applying the
reference's axis change (scene.rs:286-288).  Not part of the product."""
import base64
import json
import os
import struct

import numpy as np

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32,
         5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def load(path):
    data = open(path, "rb").read()
    buffers = []
    if data[:4] == b"glTF":
        _, _, total = struct.unpack_from("<III", data, 0)
        off = 12
        doc = None
        while off < total:
            n, kind = struct.unpack_from("<II", data, off)
            chunk = data[off + 8: off + 8 + n]
            if kind == 0x4E4F534A:
                doc = json.loads(chunk)
            elif kind == 0x004E4942:
                buffers.append(chunk)
            off += 8 + n
    else:
        doc = json.loads(data)
    bin_i = 0
    bufs = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            bufs.append(buffers[bin_i]); bin_i += 1
        elif uri.startswith("data:"):
            bufs.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            bufs.append(open(os.path.join(os.path.dirname(path), uri), "rb").read())
    return doc, bufs


def accessor(doc, bufs, idx):
    a = doc["accessors"][idx]
    bv = doc["bufferViews"][a["bufferView"]]
    dt = np.dtype(_COMP[a["componentType"]])
    nc = _NCOMP[a["type"]]
    start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    raw = bufs[bv["buffer"]]
    out = np.zeros((a["count"], nc), dtype=dt)
    for i in range(a["count"]):
        out[i] = np.frombuffer(raw, dtype=dt, count=nc, offset=start + i * stride)
    return out


def triangle_primitives(path):
    """Yields (positions[n,3] in world axes (z,x,y), indices[m]) per triangle primitive in
    document mesh order (not scene-graph order)."""
    doc, bufs = load(path)
    for mesh in doc.get("meshes", []):
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4:
                continue
            pos = accessor(doc, bufs, prim["attributes"]["POSITION"]).astype(np.float32)
            idx = accessor(doc, bufs, prim["indices"]).astype(np.uint32).reshape(-1)
            yield pos[:, [2, 0, 1]].copy(), idx
