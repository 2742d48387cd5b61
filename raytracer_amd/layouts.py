"""numpy views of the byte-exact records that cross the host<->device boundary.

Sizes and field offsets are those of the reference's pod structs (scene.rs:713-717, 103-117,
840-850, 73-78, 142-165, 181-189; input.rs:20-32; main.rs:31-43) and of their WGSL mirrors
(shader.wgsl:28-125, 162-167, 200-202, 260-292).
"""
import numpy as np

TRI = np.dtype([("v", np.float32, (3, 4))])                                   # 48 B
VERTEX_EXT = np.dtype([("tc", np.float32, (2, 2)), ("normal", np.float32, 2),
                       ("color", np.uint32), ("pad", np.float32)])            # 32 B
TRI_EXT = np.dtype([("v", VERTEX_EXT, 3)])                                    # 96 B
NODE = np.dtype([("aabb", np.float32, 6), ("first", np.uint32), ("count", np.uint32)])  # 32 B
TEX_REF = np.dtype([("offset", np.uint32), ("size", np.uint32)])              # 8 B
MATERIAL = np.dtype([
    ("albedo", TEX_REF), ("emissive", TEX_REF), ("normal", TEX_REF),
    ("metallic_roughness", TEX_REF),
    ("albedo_factor", np.float32, 4),
    ("emissive_factor", np.float32, 3),
    ("normal_scale", np.float32),
    ("albedo_texcoord", np.uint32), ("emissive_texcoord", np.uint32),
    ("normal_texcoord", np.uint32), ("metal_r_texcoord", np.uint32),
    ("metallic_factor", np.float32), ("roughness_factor", np.float32),
    ("id", np.uint32), ("alpha_settings", np.uint32),
])                                                                            # 96 B
PRIMITIVE = np.dtype([("transform", np.float32, 16), ("inv_transform", np.float32, 16),
                      ("material", MATERIAL), ("bvh_idx", np.uint32),
                      ("pad", np.uint32, 3)])                                 # 240 B
GPU_CAMERA = np.dtype([("dir", np.float32, 3), ("fovy", np.float32),
                       ("origin", np.float32, 3), ("focus", np.float32),
                       ("aperture", np.float32), ("exposure", np.float32),
                       ("bloom", np.float32), ("dispersion", np.float32)])    # 48 B
FRAME_UNIFORMS = np.dtype([
    ("lights", np.uint8, 512),
    ("camera", GPU_CAMERA),
    ("tri_count", np.uint32), ("num_point_lights", np.uint32),
    ("num_directional_lights", np.uint32), ("pad0", np.uint32),
    ("res", np.uint32, 2), ("frame", np.uint32), ("time", np.float32),
    ("reject_hist", np.uint32), ("node_count", np.uint32), ("prim_count", np.uint32),
    ("pad1", np.uint32),
])                                                                            # 608 B

assert TRI.itemsize == 48 and TRI_EXT.itemsize == 96 and NODE.itemsize == 32
assert MATERIAL.itemsize == 96 and PRIMITIVE.itemsize == 240
assert GPU_CAMERA.itemsize == 48 and FRAME_UNIFORMS.itemsize == 608

ALPHA_OPAQUE, ALPHA_MASK, ALPHA_BLEND = 0, 1, 2


def default_material():
    """gltf-crate defaults for a primitive without a material (scene.rs:451-468): base colour
    1, metallic 1, roughness 1, opaque with cutoff 0.5."""
    m = np.zeros((), dtype=MATERIAL)
    m["albedo_factor"] = (1.0, 1.0, 1.0, 1.0)
    m["normal_scale"] = 1.0
    m["metallic_factor"] = 1.0
    m["roughness_factor"] = 1.0
    m["alpha_settings"] = ALPHA_OPAQUE | (int(0.5 * 65535.0) << 16)
    return m
