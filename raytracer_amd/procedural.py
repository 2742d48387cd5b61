"""Deterministic synthetic scenes, textures and environment maps.

The reference's assets cannot travel with this repository and Khronos Sponza was never part of
it (resources/large is git-ignored upstream), so every scene used on the GPU box is generated
here with the shape of the named configuration: triangle count, primitive count, material mix
(textured, two UV sets, MASK / BLEND alpha, emissive, metallic-roughness maps), camera.
World axes are the reference's: X forward, Z up, -Y right (input.rs:6-8).
"""
import numpy as np

from . import host, layouts as L

# ------------------------------------------------------------------ meshes (positions, indices, normals, uv)


def grid(nx, ny, size=(1.0, 1.0), height=None, uv_scale=1.0):
    """(nx x ny) quads in the XY plane centred on the origin, +Z normals; height(x, y) optional."""
    xs = np.linspace(-0.5, 0.5, nx + 1) * size[0]
    ys = np.linspace(-0.5, 0.5, ny + 1) * size[1]
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    Z = np.zeros_like(X) if height is None else height(X, Y)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    if height is None:
        nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0], 1))
    else:
        dzdx = np.gradient(Z, xs, axis=0)
        dzdy = np.gradient(Z, ys, axis=1)
        nrm = np.stack([-dzdx, -dzdy, np.ones_like(Z)], -1).reshape(-1, 3).astype(np.float32)
    U, V = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="ij")
    uv = (np.stack([U, V], -1).reshape(-1, 2) * uv_scale).astype(np.float32)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    a = (i * (ny + 1) + j).reshape(-1)
    b, c, d = a + (ny + 1), a + (ny + 1) + 1, a + 1
    idx = np.stack([a, b, c, a, c, d], -1).reshape(-1).astype(np.uint32)
    return pos, idx, nrm, uv


def uv_sphere(slices, stacks, radius=1.0, displace=None):
    """Lat-long sphere with 2 * slices * (stacks - 1) triangles."""
    pos, uv = [[0, 0, 1.0]], [[0.5, 0.0]]
    for s in range(1, stacks):
        th = np.pi * s / stacks
        for k in range(slices):
            ph = 2 * np.pi * k / slices
            pos.append([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
            uv.append([k / slices, s / stacks])
    pos.append([0, 0, -1.0])
    uv.append([0.5, 1.0])
    pos = np.array(pos, np.float64)
    idx = []
    ring = lambda s, k: 1 + (s - 1) * slices + (k % slices)
    for k in range(slices):
        idx += [0, ring(1, k), ring(1, k + 1)]
    for s in range(1, stacks - 1):
        for k in range(slices):
            a, b, c, d = ring(s, k), ring(s + 1, k), ring(s + 1, k + 1), ring(s, k + 1)
            idx += [a, b, c, a, c, d]
    last = len(pos) - 1
    for k in range(slices):
        idx += [last, ring(stacks - 1, k + 1), ring(stacks - 1, k)]
    nrm = pos.copy()
    if displace is not None:
        pos = pos * displace(pos)[:, None]
    return ((pos * radius).astype(np.float32), np.array(idx, np.uint32), nrm.astype(np.float32),
            np.array(uv, np.float32))


def cylinder(segments, rings, radius=1.0, height=1.0, profile=None):
    """Open cylinder along +Z from 0 to height; profile(z01) scales the radius."""
    pos, nrm, uv = [], [], []
    for r in range(rings + 1):
        z = r / rings
        rad = radius * (1.0 if profile is None else profile(z))
        for k in range(segments + 1):
            ph = 2 * np.pi * k / segments
            pos.append([rad * np.cos(ph), rad * np.sin(ph), z * height])
            nrm.append([np.cos(ph), np.sin(ph), 0.0])
            uv.append([k / segments * 2.0, z * 4.0])
    idx = []
    for r in range(rings):
        for k in range(segments):
            a = r * (segments + 1) + k
            b, c, d = a + 1, a + segments + 2, a + segments + 1
            idx += [a, b, c, a, c, d]
    return (np.array(pos, np.float32), np.array(idx, np.uint32), np.array(nrm, np.float32),
            np.array(uv, np.float32))


def torus(seg_major, seg_minor, R=1.0, r=0.25, arc=2 * np.pi):
    pos, nrm, uv = [], [], []
    for i in range(seg_major + 1):
        a = arc * i / seg_major
        for j in range(seg_minor + 1):
            b = 2 * np.pi * j / seg_minor
            n = [np.cos(a) * np.cos(b), np.sin(a) * np.cos(b), np.sin(b)]
            pos.append([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)])
            nrm.append(n)
            uv.append([i / seg_major * 4.0, j / seg_minor])
    idx = []
    for i in range(seg_major):
        for j in range(seg_minor):
            a = i * (seg_minor + 1) + j
            b, c, d = a + seg_minor + 1, a + seg_minor + 2, a + 1
            idx += [a, b, c, a, c, d]
    return (np.array(pos, np.float32), np.array(idx, np.uint32), np.array(nrm, np.float32),
            np.array(uv, np.float32))


def box(half=(0.5, 0.5, 0.5)):
    """24 vertices, 12 triangles, per-face normals and UVs."""
    hx, hy, hz = half
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),
             ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, -1, 0), (1, 0, 0), (0, 0, 1)),
             ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    pos, nrm, uv, idx = [], [], [], []
    h = np.array([hx, hy, hz])
    for n, u, v in faces:
        n, u, v = np.array(n, float), np.array(u, float), np.array(v, float)
        base = len(pos)
        for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append((n + su * u + sv * v) * h)
            nrm.append(n)
            uv.append([(su + 1) / 2, (sv + 1) / 2])
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return (np.array(pos, np.float32), np.array(idx, np.uint32), np.array(nrm, np.float32),
            np.array(uv, np.float32))


def transform_mesh(mesh, m16):
    """Bake a column-major float[16] into a mesh (positions and normals)."""
    pos, idx, nrm, uv = mesh
    M = np.asarray(m16, np.float64).reshape(4, 4).T
    p = pos.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    n = nrm.astype(np.float64) @ np.linalg.inv(M[:3, :3])
    return p.astype(np.float32), idx, n.astype(np.float32), uv


def merge(meshes):
    pos, idx, nrm, uv, base = [], [], [], [], 0
    for p, i, n, u in meshes:
        pos.append(p); nrm.append(n); uv.append(u); idx.append(i + base)
        base += p.shape[0]
    return (np.concatenate(pos), np.concatenate(idx).astype(np.uint32), np.concatenate(nrm),
            np.concatenate(uv))


# ------------------------------------------------------------------ textures / environment


def _value_noise(h, w, cells, rng):
    """Smooth value noise in [0, 1): a (cells + 1)^2 lattice of random values, smoothstep-interpolated
    (along x first, then along y: two gathers of the full size instead of four)."""
    g = rng.random((cells + 1, cells + 1)).astype(np.float32)
    ys = np.linspace(0, cells, h, endpoint=False, dtype=np.float32)
    xs = np.linspace(0, cells, w, endpoint=False, dtype=np.float32)
    y0, x0 = ys.astype(np.intp), xs.astype(np.intp)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
    gx = g[:, x0] * (1 - fx) + g[:, x0 + 1] * fx            # (cells + 1, w)
    return gx[y0] * (1 - fy) + gx[y0 + 1] * fy


def texture_stone(size, seed, tint=(0.75, 0.7, 0.62)):
    rng = np.random.default_rng(seed)
    n = np.float32(0.5) * _value_noise(size, size, 8, rng) + np.float32(0.3) * _value_noise(size, size, 32, rng) + \
        np.float32(0.2) * _value_noise(size, size, 128, rng)
    yy, xx = np.arange(size)[:, None], np.arange(size)[None, :]
    mortar = ((yy % (size // 8) < 3) | ((xx + (yy // (size // 8)) * (size // 16)) % (size // 4) < 3))
    v = np.clip(np.float32(0.35) + np.float32(0.65) * n, 0, 1) * np.where(mortar, np.float32(0.55), np.float32(1.0))
    img = np.empty((size, size, 4), np.uint8)
    for k in range(3):
        img[..., k] = np.clip(v * np.float32(tint[k] * 255), 0, 255)
    img[..., 3] = 255
    return img


def texture_fabric(size, seed, tint):
    rng = np.random.default_rng(seed)
    yy, xx = np.arange(size, dtype=np.float32)[:, None], np.arange(size, dtype=np.float32)[None, :]
    weave = np.float32(0.5) + np.float32(0.25) * np.sin(xx * np.float32(2 * np.pi / 8)) * np.sin(yy * np.float32(2 * np.pi / 8))
    stripes = np.float32(0.75) + np.float32(0.25) * np.sign(np.sin(xx * np.float32(2 * np.pi / (size / 6))))
    v = weave * stripes * (np.float32(0.8) + np.float32(0.2) * _value_noise(size, size, 16, rng))
    img = np.zeros((size, size, 4), np.uint8)
    for k in range(3):
        img[..., k] = np.clip(v * np.float32(tint[k] * 255), 0, 255)
    img[..., 3] = 255
    return img


def texture_leaves(size, seed):
    """Foliage atlas with a binary-ish alpha channel (for MASK) and soft edges (for BLEND)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.arange(size, dtype=np.float32)[:, None], np.arange(size, dtype=np.float32)[None, :]
    alpha = np.zeros((size, size), np.float32)
    for _ in range(24):
        cx, cy = (np.float32(c) for c in rng.random(2) * size)
        a, b = np.float32(size * (0.05 + 0.08 * rng.random())), np.float32(size * (0.02 + 0.04 * rng.random()))
        th = rng.random() * np.pi
        cs, sn = np.float32(np.cos(th)), np.float32(np.sin(th))
        dx, dy = xx - cx, yy - cy
        u, v = dx * cs + dy * sn, dy * cs - dx * sn
        alpha = np.maximum(alpha, np.clip(np.float32(1.5) - np.float32(1.5) * ((u / a) ** 2 + (v / b) ** 2), 0, 1))
    g = 0.35 + 0.4 * _value_noise(size, size, 16, rng)
    img = np.zeros((size, size, 4), np.uint8)
    img[..., 0] = np.clip(g * 0.45 * 255, 0, 255)
    img[..., 1] = np.clip(g * 255, 0, 255)
    img[..., 2] = np.clip(g * 0.3 * 255, 0, 255)
    img[..., 3] = np.clip(alpha * 255, 0, 255)
    return img


def texture_metal_rough(size, seed):
    """glTF metallic-roughness map: G = roughness, B = metallic."""
    rng = np.random.default_rng(seed)
    img = np.zeros((size, size, 4), np.uint8)
    img[..., 1] = np.clip((0.2 + 0.6 * _value_noise(size, size, 8, rng)) * 255, 0, 255)
    img[..., 2] = np.where(_value_noise(size, size, 4, rng) > 0.5, 255, 0)
    img[..., 3] = 255
    return img


def texture_surface_mr(size, seed, roughness=(0.35, 0.95), metal=0.0):
    """glTF metallic-roughness map of a dielectric or metal surface: G = roughness varying between
    the two bounds, B = metallic: 0, 255, or (0 < metal < 1) patches covering about that fraction."""
    rng = np.random.default_rng(seed)
    n = 0.6 * _value_noise(size, size, 6, rng) + 0.4 * _value_noise(size, size, 48, rng)
    img = np.zeros((size, size, 4), np.uint8)
    img[..., 1] = np.clip((roughness[0] + (roughness[1] - roughness[0]) * n) * 255, 0, 255)
    if metal >= 1.0:
        img[..., 2] = 255
    elif metal > 0.0:
        img[..., 2] = np.where(_value_noise(size, size, 5, rng) < metal, 255, 0)
    img[..., 3] = 255
    return img


def texture_normal(size, seed, strength=2.0):
    """Tangent-space normal map from the slopes of a noise height field.  The reference loads a
    material's normal texture into the atlas (scene.rs:290-341) and never samples it
    (shader.wgsl:803-881 reads albedo, emissive and metallic-roughness only): it is there for its bytes."""
    rng = np.random.default_rng(seed)
    h = 0.7 * _value_noise(size, size, 12, rng) + 0.3 * _value_noise(size, size, 64, rng)
    gx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * strength * size / 64.0
    gy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * strength * size / 64.0
    nz = 1.0 / np.sqrt(gx * gx + gy * gy + 1.0)
    img = np.zeros((size, size, 4), np.uint8)
    img[..., 0] = np.clip((-gx * nz * 0.5 + 0.5) * 255, 0, 255)
    img[..., 1] = np.clip((-gy * nz * 0.5 + 0.5) * 255, 0, 255)
    img[..., 2] = np.clip((nz * 0.5 + 0.5) * 255, 0, 255)
    img[..., 3] = 255
    return img


def texture_checker(size, cells=8, a=(230, 230, 230, 255), b=(40, 40, 40, 255)):
    yy, xx = np.mgrid[0:size, 0:size]
    m = ((xx * cells // size) + (yy * cells // size)) % 2 == 0
    img = np.where(m[..., None], np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)
    return img


def sky_env(w=1024, h=512, sun_dir=(0.31, 0.25, 0.92), sun_power=60.0):
    """Equirectangular rgba32f sky with a bright sun lobe, laid out as sample_env_map reads it
    (shader.wgsl:691-694): u = atan2(y, x) / 2pi + 0.5, v = asin(-z) / pi + 0.5."""
    u = (np.arange(w) + 0.5) / w
    v = (np.arange(h) + 0.5) / h
    phi = (u - 0.5) * 2 * np.pi
    z = -np.sin((v - 0.5) * np.pi)
    c = np.sqrt(np.maximum(1 - z * z, 0))
    d = np.stack([c[:, None] * np.cos(phi)[None, :], c[:, None] * np.sin(phi)[None, :],
                  np.repeat(z[:, None], w, 1)], -1)
    s = np.array(sun_dir, float)
    s /= np.linalg.norm(s)
    cosang = np.clip(d @ s, -1, 1)
    up = np.clip(d[..., 2], 0, 1)
    sky = np.stack([0.25 + 0.25 * (1 - up), 0.4 + 0.3 * (1 - up), 0.9 - 0.1 * up], -1) * (0.6 + 0.6 * up[..., None])
    ground = np.array([0.18, 0.16, 0.13])
    col = np.where(d[..., 2:3] >= 0, sky, ground)
    col = col + sun_power * np.array([1.0, 0.85, 0.6]) * (np.clip(cosang, 0, 1) ** 600)[..., None]
    col = col + 1.5 * np.array([1.0, 0.8, 0.6]) * (np.clip(cosang, 0, 1) ** 12)[..., None]
    env = np.ones((h, w, 4), np.float32)
    env[..., :3] = col
    return env


# ------------------------------------------------------------------ materials


def material(albedo=(1, 1, 1, 1), metallic=0.0, roughness=1.0, emissive=(0, 0, 0), albedo_tex=None,
             emissive_tex=None, mr_tex=None, albedo_set=0, emissive_set=0, mr_set=0,
             alpha_mode=L.ALPHA_OPAQUE, alpha_cutoff=0.5, normal_tex=None):
    m = L.default_material()
    m["albedo_factor"] = albedo
    m["metallic_factor"], m["roughness_factor"] = metallic, roughness
    m["emissive_factor"] = emissive
    if albedo_tex is not None:
        m["albedo"] = albedo_tex
    if emissive_tex is not None:
        m["emissive"] = emissive_tex
    if mr_tex is not None:
        m["metallic_roughness"] = mr_tex
    if normal_tex is not None:
        m["normal"] = normal_tex
    m["albedo_texcoord"], m["emissive_texcoord"], m["metal_r_texcoord"] = albedo_set, emissive_set, mr_set
    m["alpha_settings"] = host.pack_alpha(alpha_mode, alpha_cutoff)
    return m


# ------------------------------------------------------------------ scenes


def box_scene(env_size=(256, 128)):
    """Stand-in for resources/Box.gltf: a unit cube (12 triangles, one primitive), base colour
    (0.8, 0, 0), metallic 0, roughness 1, default camera (input.rs:202-219)."""
    s = host.Scene()
    pos, idx, nrm, uv = box()
    s.add_primitive(pos, idx, normals=nrm, material=material(albedo=(0.8, 0, 0, 1)))
    s.set_env(sky_env(*env_size))
    s.add_camera(host.Camera.default())
    return s


def blob_scene(slices=96, stacks=83, env_size=(1024, 512)):
    """Stand-in for resources/suzanne.glb: one primitive of 15 744 triangles without material
    (gltf defaults: metallic 1, roughness 1), no UVs, default camera."""
    def bumps(p):
        return (1.0 + 0.18 * np.sin(5 * p[:, 0]) * np.cos(4 * p[:, 1]) + 0.12 * np.sin(7 * p[:, 2] + 1.0)
                + 0.25 * np.exp(-8 * ((p[:, 0] - 0.7) ** 2 + (np.abs(p[:, 1]) - 0.45) ** 2 + (p[:, 2] - 0.3) ** 2)))
    pos, idx, nrm, _ = uv_sphere(slices, stacks, radius=1.6, displace=bumps)
    s = host.Scene()
    s.add_primitive(pos, idx, normals=nrm)
    s.set_env(sky_env(*env_size))
    s.add_camera(host.Camera.default())
    return s


def material_test_scene(tex=64, env_size=(256, 128)):
    """Small scene that reaches every material branch of the path: albedo texture on UV set 1,
    emissive texture, metallic-roughness texture, MASK and BLEND alpha, an instanced primitive
    with a non-uniform scale, and a primitive without normals."""
    s = host.Scene()
    stone = s.add_texture(texture_stone(tex, 1))
    leaves = s.add_texture(texture_leaves(tex, 2))
    mr = s.add_texture(texture_metal_rough(tex, 3))
    glow = s.add_texture(texture_checker(tex, 4, (255, 180, 60, 255), (10, 10, 30, 255)))

    floor = grid(24, 24, (14.0, 14.0), uv_scale=6.0)
    s.add_primitive(floor[0], floor[1], normals=floor[2], uv0=floor[3] * 0.0, uv1=floor[3],
                    material=material(albedo_tex=stone, albedo_set=1, roughness=0.9))
    sph = uv_sphere(24, 16, radius=1.0)
    p_sph = s.add_primitive(sph[0], sph[1], normals=sph[2], uv0=sph[3],
                            material=material(albedo=(0.9, 0.85, 0.8, 1), mr_tex=mr, roughness=1.0, metallic=1.0),
                            transform=host.trs((0.5, -1.8, 1.0)))
    s.add_instance(p_sph, host.trs((1.5, 2.0, 0.7), rot_z=0.6, scale=(0.7, 1.3, 0.6)))
    card = grid(6, 6, (3.0, 3.0), uv_scale=1.0)
    s.add_primitive(card[0], card[1], normals=card[2], uv0=card[3],
                    material=material(albedo_tex=leaves, alpha_mode=L.ALPHA_MASK, alpha_cutoff=0.5, roughness=0.8),
                    transform=host.trs((-1.0, 0.0, 1.6), rot_y=np.pi / 2))
    s.add_primitive(card[0], card[1], normals=card[2], uv0=card[3],
                    material=material(albedo_tex=leaves, alpha_mode=L.ALPHA_BLEND, roughness=0.8),
                    transform=host.trs((-2.2, 0.4, 1.6), rot_y=np.pi / 2, rot_z=0.3))
    lamp = box((0.4, 0.4, 0.05))
    s.add_primitive(lamp[0], lamp[1], normals=lamp[2], uv0=lamp[3],
                    material=material(albedo=(0.2, 0.2, 0.2, 1), emissive=(6.0, 6.0, 6.0), emissive_tex=glow),
                    transform=host.trs((0.5, 0.3, 3.2)))
    bare = box((0.5, 0.5, 0.5))
    s.add_primitive(bare[0], bare[1], material=material(albedo=(0.3, 0.5, 0.9, 1), roughness=0.3),
                    transform=host.trs((2.5, -0.5, 0.5), rot_z=0.4))
    s.set_env(sky_env(*env_size))
    s.add_camera(host.Camera.look((-6.0, 0.5, 3.0), yaw=-0.05, pitch=np.radians(-17), fovy=np.radians(60), focus=6.0))
    return s


def terrain_mesh(n=60, size=40.0, seed=7):
    rng = np.random.default_rng(seed)
    k = rng.random((4, 3)) * 2 + 0.5

    def height(X, Y):
        z = np.zeros_like(X)
        for a, b, c in k:
            z += 0.9 / a * np.sin(a * X * 0.35 + c) * np.cos(b * Y * 0.3 + a)
        return z
    return grid(n, n, (size, size), height=height, uv_scale=8.0)


def terrain_scene(tiles=1, n=60, env_size=(1024, 512), instanced=False, view="ground", shuffled=False):
    """Stand-in for resources/simple_terrain.glb (a ~7 k-triangle plane plus a ~1 k-triangle
    object, own camera) and, with tiles > 1, for its tiling into one large primitive
    (tiles^2 copies merged into one BVH): the HBM-resident traversal stress case.  With
    instanced=True the tiles stay 2 * tiles^2 primitives that share two BVHs: the stress case
    for the number of primitives.  view="ground" is the glTF-style camera near the first tiles
    (it sees a sliver of a large tiling); view="top" looks down on the whole tiling from a height at
    which the image is as wide as the terrain, so that every pixel lands in another part of it.
    shuffled=True lists the instances in random instead of row-major order: neighbours in the
    primitive list are then no neighbours in space (the worst case for bounds over index ranges)."""
    size = 40.0
    t = terrain_mesh(n, size)
    rock = uv_sphere(22, 23, radius=2.0)  # 968 triangles
    s = host.Scene()
    if tiles == 1:
        s.add_primitive(t[0], t[1], normals=t[2], uv0=t[3], material=material(albedo=(0.8, 0.8, 0.8, 1), roughness=1.0))
        s.add_primitive(rock[0], rock[1], normals=rock[2], uv0=rock[3],
                        material=material(albedo=(0.8, 0.8, 0.8, 1), roughness=1.0), transform=host.trs((0.0, 0.0, 2.5)))
    elif instanced:
        mat = material(albedo=(0.8, 0.8, 0.8, 1), roughness=1.0)
        pos = lambda i, j, z: host.trs(((i - tiles / 2 + 0.5) * size, (j - tiles / 2 + 0.5) * size, z))
        pt = s.add_primitive(t[0], t[1], normals=t[2], uv0=t[3], material=mat, transform=pos(0, 0, 0.0))
        pr = s.add_primitive(rock[0], rock[1], normals=rock[2], uv0=rock[3], material=mat, transform=pos(0, 0, 2.5))
        cells = [(i, j) for i in range(tiles) for j in range(tiles) if i or j]
        if shuffled:
            cells = [cells[k] for k in np.random.default_rng(tiles).permutation(len(cells))]
        for i, j in cells:
            s.add_instance(pt, pos(i, j, 0.0))
            s.add_instance(pr, pos(i, j, 2.5))
    else:
        parts = []
        for i in range(tiles):
            for j in range(tiles):
                off = host.trs(((i - tiles / 2 + 0.5) * size, (j - tiles / 2 + 0.5) * size, 0.0))
                parts.append(transform_mesh(t, off))
                parts.append(transform_mesh(rock, host.trs(((i - tiles / 2 + 0.5) * size, (j - tiles / 2 + 0.5) * size, 2.5))))
        m = merge(parts)
        s.add_primitive(m[0], m[1], normals=m[2], uv0=m[3], material=material(albedo=(0.8, 0.8, 0.8, 1), roughness=1.0))
    s.set_env(sky_env(*env_size))
    if view == "top":
        fovy = np.radians(60.0)
        height = (tiles * size / 2) / (np.tan(fovy / 2) * 16.0 / 9.0)   # 16:9: the width of the image spans the tiling
        # camera_ray's basis needs forward x (0,0,1) != 0 (shader.wgsl:746): not quite straight down
        s.add_camera(host.Camera.look((-height * np.tan(np.radians(5.0)), 0.0, height), yaw=0.0, pitch=np.radians(-85.0),
                                      fovy=fovy, focus=height))
    else:
        s.add_camera(host.Camera.look((-38.0, -6.0, 11.0), yaw=0.15, pitch=np.radians(-14), fovy=0.3562, focus=50.0))
    return s


def atrium_scene(detail=1.0, tex=512, env_size=(1024, 512), textures="sponza"):
    """Sponza-shaped stand-in: a two-storey colonnaded atrium open to the sky, ~262 k triangles
    in 103 primitives (detail = 1), stone / fabric / foliage textures, MASK alpha on foliage and
    chains, instanced columns, camera on the long axis with depth of field.

    textures="sponza" (default) dresses it the way the glTF Sponza is dressed: 31 materials, every
    one with an albedo map, a metallic-roughness map and a normal map of its own (93 textures of
    tex x tex texels in the one rgba8 atlas of scene.rs:290-341: 390 MB at tex = 1024; the normal
    maps are loaded and never sampled, as upstream).  textures="light" is the earlier, lighter
    dressing (10 textures shared by all materials, a metallic-roughness map on the columns and the
    basin only), kept so that both can be measured side by side."""
    assert textures in ("sponza", "light")
    heavy = textures == "sponza"
    s = host.Scene()
    d = lambda n: max(2, int(round(n * np.sqrt(detail))))
    stone_tints = [(0.75, 0.7, 0.62), (0.6, 0.58, 0.55), (0.8, 0.72, 0.6), (0.55, 0.5, 0.45)]
    fabric_tints = [(0.8, 0.15, 0.12), (0.15, 0.3, 0.75), (0.2, 0.6, 0.25)]
    if not heavy:
        stone = [s.add_texture(texture_stone(tex, 10 + k, tint)) for k, tint in enumerate(stone_tints)]
        fabric = [s.add_texture(texture_fabric(tex, 20 + k, tint)) for k, tint in enumerate(fabric_tints)]
        leaves = s.add_texture(texture_leaves(tex, 30))
        chain_tex = s.add_texture(texture_checker(tex // 2, 16, (200, 190, 160, 255), (0, 0, 0, 0)))
        mr = s.add_texture(texture_metal_rough(tex // 2, 31))
    made = {}

    def mat(kind, i, light, **kw):
        """Material i of a kind.  Heavy dressing: its own albedo, metallic-roughness and normal map,
        made on first use; light dressing: the material the scene used before (light())."""
        if not heavy:
            return light()
        key = (kind, i)
        if key not in made:
            seed = 1000 + 37 * len(made)
            if kind == "stone":
                t = stone_tints[i % 4]
                shade = 0.85 + 0.05 * (i % 7)
                alb = texture_stone(tex, seed, (min(t[0] * shade, 1.0), min(t[1] * shade, 1.0), min(t[2] * shade, 1.0)))
                mrt = texture_surface_mr(tex, seed + 1, (0.45, 1.0), 0.0)
            elif kind == "fabric":
                t = fabric_tints[i % 3]
                alb = texture_fabric(tex, seed, t if i < 3 else (t[1], t[2], t[0]))
                mrt = texture_surface_mr(tex, seed + 1, (0.8, 1.0), 0.0)
            elif kind == "leaves":
                alb = texture_leaves(tex, seed)
                mrt = texture_surface_mr(tex, seed + 1, (0.5, 0.9), 0.0)
            elif kind == "chain":
                alb = np.kron(texture_checker(tex // 16, 16, (200, 190, 160, 255), (0, 0, 0, 0)), np.ones((16, 16, 1), np.uint8))
                mrt = texture_surface_mr(tex, seed + 1, (0.25, 0.6), 1.0)
            elif kind == "metal":
                alb = texture_stone(tex, seed, (0.95, 0.78, 0.34))
                mrt = texture_surface_mr(tex, seed + 1, (0.2, 0.5), 1.0)
            elif kind == "glazed":
                alb = texture_stone(tex, seed, stone_tints[i % 4])
                mrt = texture_surface_mr(tex, seed + 1, (0.3, 0.8), 0.5)
            else:
                raise ValueError(kind)
            made[key] = (s.add_texture(alb), s.add_texture(mrt), s.add_texture(texture_normal(tex, seed + 2)))
        alb, mrt, nrm = made[key]
        return material(albedo_tex=alb, mr_tex=mrt, normal_tex=nrm, **kw)

    L_, W_, H1, H2 = 30.0, 12.0, 5.0, 10.0
    # floor and upper gallery floors
    f = grid(d(186), d(72), (L_, W_), uv_scale=10.0)
    s.add_primitive(f[0], f[1], normals=f[2], uv0=f[3],
                    material=mat("stone", 0, lambda: material(albedo_tex=stone[0], roughness=0.6), roughness=0.75))
    for k, side in enumerate((-1, 1)):
        g = grid(d(96), d(12), (L_, 3.0), uv_scale=6.0)
        s.add_primitive(g[0], g[1], normals=g[2], uv0=g[3],
                        material=mat("stone", 1 + k, lambda: material(albedo_tex=stone[1], roughness=0.8), roughness=0.9),
                        transform=host.trs((0.0, side * (W_ / 2 - 1.5), H1)))
    # outer walls (4), two storeys
    for k, side in enumerate((-1, 1)):
        w = grid(d(110), d(40), (L_, H2), uv_scale=8.0)
        s.add_primitive(w[0], w[1], normals=w[2], uv0=w[3],
                        material=mat("stone", 3 + k, lambda: material(albedo_tex=stone[2], roughness=0.9), roughness=1.0),
                        transform=host.trs((0.0, side * W_ / 2, H2 / 2), rot_x=side * np.pi / 2))
        e = grid(d(48), d(40), (W_, H2), uv_scale=8.0)
        s.add_primitive(e[0], e[1], normals=e[2], uv0=e[3],
                        material=mat("stone", 5 + k, lambda: material(albedo_tex=stone[3], roughness=0.9), roughness=1.0),
                        transform=host.basis((side * L_ / 2, 0.0, H2 / 2), (0, 1, 0), (0, 0, 1), (1, 0, 0)))
    # columns: one unique mesh per storey and side, instanced along the nave
    flute = lambda z: 1.0 + 0.25 * np.exp(-40 * z * z) + 0.3 * np.exp(-40 * (1 - z) ** 2) - 0.06 * z
    n_cols = 9
    for storey, (z0, hh, rad) in enumerate(((0.0, H1, 0.38), (H1, H2 - H1 - 0.5, 0.3))):
        for k, side in enumerate((-1, 1)):
            c = cylinder(d(48), d(40), rad, hh, flute)
            y = side * (W_ / 2 - 3.0)
            first = s.add_primitive(c[0], c[1], normals=c[2], uv0=c[3],
                                    material=mat("glazed", 2 * storey + k,
                                                 lambda: material(albedo_tex=stone[(storey + 1) % 4], mr_tex=mr, roughness=0.7), roughness=0.9),
                                    transform=host.trs((-L_ / 2 + 2.0, y, z0)))
            for k in range(1, n_cols):
                s.add_instance(first, host.trs((-L_ / 2 + 2.0 + k * (L_ - 4.0) / (n_cols - 1), y, z0), rot_z=0.37 * k))
    # arches between ground-floor columns
    for side in (-1, 1):
        for k in range(n_cols - 1):
            a = torus(d(28), d(12), R=(L_ - 4.0) / (n_cols - 1) / 2, r=0.22, arc=np.pi)
            x = -L_ / 2 + 2.0 + (k + 0.5) * (L_ - 4.0) / (n_cols - 1)
            s.add_primitive(a[0], a[1], normals=a[2], uv0=a[3],
                            material=mat("stone", 7 + k % 4, lambda: material(albedo_tex=stone[k % 4], roughness=0.8), roughness=0.9),
                            transform=host.trs((x, side * (W_ / 2 - 3.0), H1 - 1.7), rot_x=np.pi / 2))
    # hanging curtains (wavy cloth), three fabrics
    for k in range(6):
        wave = lambda X, Y, ph=k: 0.18 * np.sin(6 * X + ph) * (0.4 + 0.6 * (Y + 1.5) / 3.0)
        cl = grid(d(84), d(84), (3.2, 3.0), height=wave, uv_scale=2.0)
        side = -1 if k % 2 else 1
        s.add_primitive(cl[0], cl[1], normals=cl[2], uv0=cl[3], uv1=cl[3] * 3.0,
                        material=mat("fabric", k, lambda: material(albedo_tex=fabric[k % 3], albedo_set=k % 2, roughness=0.95),
                                     albedo_set=k % 2, mr_set=k % 2, roughness=1.0),
                        transform=host.trs((-9.0 + 3.6 * k, side * (W_ / 2 - 3.0), H1 + 1.8), rot_x=np.pi / 2))
    # potted plants: MASK-alpha leaf cards
    rng = np.random.default_rng(5)
    for k in range(8):
        cards = []
        for _ in range(int(90 * detail) + 4):
            q = grid(2, 2, (0.9, 0.5))
            m = host.trs((rng.normal(0, 0.35), rng.normal(0, 0.35), 0.9 + rng.random() * 1.2),
                         rot_z=rng.random() * 6.28, rot_y=rng.normal(0, 0.8), rot_x=rng.normal(0, 0.8))
            cards.append(transform_mesh(q, m))
        pm = merge(cards)
        x = -11.0 + 3.1 * k
        s.add_primitive(pm[0], pm[1], normals=pm[2], uv0=pm[3],
                        material=mat("leaves", k % 2, lambda: material(albedo_tex=leaves, alpha_mode=L.ALPHA_MASK, alpha_cutoff=0.5, roughness=0.7),
                                     alpha_mode=L.ALPHA_MASK, alpha_cutoff=0.5, roughness=0.9),
                        transform=host.trs((x, (1 if k % 2 else -1) * 1.6, 0.0)))
        pot = cylinder(d(24), d(6), 0.45, 0.9, lambda z: 0.7 + 0.3 * z)
        s.add_primitive(pot[0], pot[1], normals=pot[2], uv0=pot[3],
                        material=mat("glazed", 4, lambda: material(albedo_tex=stone[3], roughness=0.5), roughness=0.6),
                        transform=host.trs((x, (1 if k % 2 else -1) * 1.6, 0.0)))
    # sculpted heads at both ends (dense meshes)
    for k, side in enumerate((-1, 1)):
        bump = lambda p: 1.0 + 0.12 * np.sin(9 * p[:, 0]) * np.sin(8 * p[:, 1]) * np.sin(7 * p[:, 2]) + 0.2 * np.exp(-6 * (p[:, 0] - side * 0.8) ** 2)
        hd = uv_sphere(d(160), d(118), radius=1.3, displace=bump)
        s.add_primitive(hd[0], hd[1], normals=hd[2], uv0=hd[3],
                        material=mat("metal", k, lambda: material(albedo=(0.85, 0.7, 0.3, 1), metallic=1.0, roughness=0.35), roughness=0.8),
                        transform=host.trs((side * (L_ / 2 - 2.0), 0.0, 2.2)))
        ped = box((0.9, 0.9, 0.5))
        s.add_primitive(ped[0], ped[1], normals=ped[2], uv0=ped[3],
                        material=mat("stone", 11, lambda: material(albedo_tex=stone[1], roughness=0.6), roughness=0.7),
                        transform=host.trs((side * (L_ / 2 - 2.0), 0.0, 0.5)))
    # chains with MASK alpha and hanging lamps (emissive)
    for k in range(4):
        ch = cylinder(d(12), d(40), 0.05, 3.0)
        x = -9.0 + 6.0 * k
        s.add_primitive(ch[0], ch[1], normals=ch[2], uv0=ch[3] * np.array([1.0, 6.0], np.float32),
                        material=mat("chain", 0, lambda: material(albedo_tex=chain_tex, alpha_mode=L.ALPHA_MASK, alpha_cutoff=0.5, metallic=1.0, roughness=0.4),
                                     alpha_mode=L.ALPHA_MASK, alpha_cutoff=0.5, roughness=1.0),
                        transform=host.trs((x, 0.0, H2 - 3.5)))
        lamp = uv_sphere(d(24), d(16), radius=0.35)
        s.add_primitive(lamp[0], lamp[1], normals=lamp[2], uv0=lamp[3],
                        material=mat("stone", 12, lambda: material(albedo=(0.9, 0.9, 0.9, 1), emissive=(8.0, 6.0, 3.5), roughness=0.9),
                                     emissive=(8.0, 6.0, 3.5), roughness=1.0),
                        transform=host.trs((x, 0.0, H2 - 3.8)))
    # central basin
    bs = torus(d(96), d(24), R=1.6, r=0.35)
    s.add_primitive(bs[0], bs[1], normals=bs[2], uv0=bs[3],
                    material=mat("glazed", 5, lambda: material(albedo_tex=stone[0], mr_tex=mr, roughness=0.5), roughness=0.7),
                    transform=host.trs((0.0, 0.0, 0.35)))
    # roof beams across the opening
    for k in range(9):
        bm = box((0.15, W_ / 2, 0.2))
        s.add_primitive(bm[0], bm[1], normals=bm[2], uv0=bm[3],
                        material=mat("stone", 13, lambda: material(albedo_tex=stone[2], roughness=0.9), roughness=1.0),
                        transform=host.trs((-L_ / 2 + 3.0 + 3.0 * k, 0.0, H2 + 0.2)))
    s.set_env(sky_env(*env_size))
    s.add_camera(host.Camera.look((-9.5, -0.6, 1.9), yaw=0.04, pitch=np.radians(6), fovy=np.radians(62), focus=9.0))
    return s
