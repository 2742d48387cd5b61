"""Builds small .glb files in memory for the loader tests (test helper, not product)."""
import io
import json
import struct
import zlib

import numpy as np


def png_bytes(img):
    """Minimal PNG encoder for (h, w, 4) uint8 RGBA, filter type 0."""
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


class Glb:
    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": []}], "scene": 0, "nodes": [], "meshes": [],
                    "accessors": [], "bufferViews": [], "buffers": [{}], "materials": [], "textures": [], "images": [],
                    "cameras": []}

    def _view(self, data):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += data
        self.doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, type_, comp, normalized=False):
        arr = np.ascontiguousarray(arr)
        a = {"bufferView": self._view(arr.tobytes()), "componentType": comp, "count": int(arr.shape[0]), "type": type_}
        if normalized:
            a["normalized"] = True
        if type_ == "VEC3" and comp == 5126:
            a["min"], a["max"] = arr.min(0).tolist(), arr.max(0).tolist()
        self.doc["accessors"].append(a)
        return len(self.doc["accessors"]) - 1

    def image_texture(self, rgba):
        self.doc["images"].append({"bufferView": self._view(png_bytes(rgba)), "mimeType": "image/png"})
        self.doc["textures"].append({"source": len(self.doc["images"]) - 1})
        return len(self.doc["textures"]) - 1

    def material(self, **kw):
        self.doc["materials"].append(kw)
        return len(self.doc["materials"]) - 1

    def mesh(self, prims):
        self.doc["meshes"].append({"primitives": prims})
        return len(self.doc["meshes"]) - 1

    def node(self, root=True, **kw):
        self.doc["nodes"].append(kw)
        i = len(self.doc["nodes"]) - 1
        if root:
            self.doc["scenes"][0]["nodes"].append(i)
        return i

    def tobytes(self):
        doc = {k: v for k, v in self.doc.items() if not (isinstance(v, list) and not v)}
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        b = bytes(self.bin) + b"\x00" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(b)
        return (struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js +
                struct.pack("<II", len(b), 0x004E4942) + b)
