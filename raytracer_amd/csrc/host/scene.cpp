// Scene assembly, cameras, frame uniforms and the C ABI of the host-side scene pipeline.
// Stands where the reference's Scene (scene.rs:223-710) and Camera (input.rs:34-254) stand and
// produces the same flat arrays, byte for byte, from the same inputs.

#include "rthost_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <limits>

namespace rthost {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

// ---------------------------------------------------------------- small linear algebra
// glam's scalar Mat4 * Mat4: each result column is ((x*c.x + y*c.y) + z*c.z) + w*c.w.
void mat4_mul(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++)
            r[c * 4 + row] = ((a[row] * b[c * 4] + a[4 + row] * b[c * 4 + 1]) + a[8 + row] * b[c * 4 + 2]) +
                             a[12 + row] * b[c * 4 + 3];
    std::memcpy(out, r, sizeof r);
}

// input.rs:67-76: change of basis glTF (Y up, Z forward) -> world (Z up, X forward)
void from_gltf_mat4(const float gltf[16], float out[16]) {
    static const float C[16] = {0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1};
    static const float Ct[16] = {0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1};
    float tmp[16];
    mat4_mul(gltf, Ct, tmp);
    mat4_mul(C, tmp, out);
}

// 4x4 inverse by 2x2 sub-determinant factors (the formulation glam's Mat4/DMat4::inverse use).
template <typename T>
static void inverse_t(const T m[16], T out[16]) {
    const T m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    const T m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const T m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    const T m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];

    const T coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const T coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const T coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const T coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const T coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const T coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;

    const T fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
    const T fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
    const T fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    const T vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01};
    const T vec2[4] = {m12, m02, m02, m02}, vec3[4] = {m13, m03, m03, m03};
    const T sign_a[4] = {1, -1, 1, -1}, sign_b[4] = {-1, 1, -1, 1};

    T inv[16];
    for (int k = 0; k < 4; k++) {
        inv[0 + k] = ((vec1[k] * fac0[k] - vec2[k] * fac1[k]) + vec3[k] * fac2[k]) * sign_a[k];
        inv[4 + k] = ((vec0[k] * fac0[k] - vec2[k] * fac3[k]) + vec3[k] * fac4[k]) * sign_b[k];
        inv[8 + k] = ((vec0[k] * fac1[k] - vec1[k] * fac3[k]) + vec3[k] * fac5[k]) * sign_a[k];
        inv[12 + k] = ((vec0[k] * fac2[k] - vec1[k] * fac4[k]) + vec2[k] * fac5[k]) * sign_b[k];
    }
    const T d0 = m00 * inv[0], d1 = m01 * inv[4], d2 = m02 * inv[8], d3 = m03 * inv[12];
    const T det = ((d0 + d1) + d2) + d3;
    const T rcp = T(1) / det;
    for (int k = 0; k < 16; k++) out[k] = inv[k] * rcp;
}

void mat4_inverse_f64(const float m[16], float out[16]) { // scene.rs:195
    double d[16], r[16];
    for (int k = 0; k < 16; k++) d[k] = (double)m[k];
    inverse_t<double>(d, r);
    for (int k = 0; k < 16; k++) out[k] = (float)r[k];
}
void mat4_inverse_f32(const float m[16], float out[16]) { // scene.rs:382, 397
    float r[16];
    inverse_t<float>(m, r);
    std::memcpy(out, r, sizeof r);
}

static inline float signum(float x) { return std::signbit(x) ? -1.0f : 1.0f; }

void pack_oct(const float n_in[3], float out[2]) { // scene.rs:203-221
    const float s = (std::fabs(n_in[0]) + std::fabs(n_in[1])) + std::fabs(n_in[2]);
    float x = n_in[0] / s, y = n_in[1] / s;
    const float z = n_in[2] / s;
    if (z < 0.0f) {
        const float wx = (1.0f - std::fabs(y)) * signum(x);
        const float wy = (1.0f - std::fabs(x)) * signum(y);
        x = wx;
        y = wy;
    }
    out[0] = x * 0.5f + 0.5f;
    out[1] = y * 0.5f + 0.5f;
}

uint32_t pack_alpha(uint32_t mode, float cutoff) { // scene.rs:170-177
    const float c = std::min(std::max(cutoff, 0.0f), 1.0f);
    return mode | ((uint32_t)(c * 65535.0f) << 16);
}

static void tri_new(const float* p1, const float* p2, const float* p3, Tri& t) { // scene.rs:720-725
    for (int k = 0; k < 3; k++) {
        t.v[0][k] = p1[k];
        t.v[1][k] = p2[k];
        t.v[2][k] = p3[k];
        t.v[k][3] = ((p1[k] + p2[k]) + p3[k]) / 3.0f;
    }
}

// ---------------------------------------------------------------- scene assembly

int add_texture(Scene& s, const uint8_t* rgba8, uint32_t w, uint32_t h, TexRef* out) {
    if (!rgba8 || w == 0 || h == 0) { set_error("add_texture: empty image"); return RTHOST_ERR_INVALID; }
    if (w > 0xFFFFu || h > 0xFFFFu) { // scene.rs:82 packs w<<16 | h
        set_error("add_texture: dimension above 65535 does not fit the packed size word");
        return RTHOST_ERR_LIMIT;
    }
    const uint64_t n = (uint64_t)w * h;
    if (s.texels.size() + n > 0xFFFFFFFFull) { set_error("add_texture: texel offset overflows u32"); return RTHOST_ERR_LIMIT; }
    out->offset = (uint32_t)s.texels.size();
    out->size = (w << 16) | h;
    // the atlas grows geometrically: an exact reserve per texture would copy it once per texture
    const size_t at = s.texels.size();
    if (s.texels.capacity() < at + n) s.texels.reserve(std::max<size_t>(at + n, s.texels.capacity() * 2));
    s.texels.resize(at + n);
    uint32_t* dst = s.texels.data() + at;
    for (uint64_t i = 0; i < n; i++) { // scene.rs:290-297
        const uint8_t* p = rgba8 + i * 4;
        dst[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    }
    return RTHOST_OK;
}

int add_primitive(Scene& s, const float* positions, uint32_t n_verts, const uint32_t* indices,
                  uint64_t n_indices, const float* normals, const float* uv0, const float* uv1,
                  const uint32_t* colors, const Material& material, const float transform[16],
                  uint32_t* out_prim, bool normals_are_unit) {
    if (!positions || !indices || !transform) { set_error("add_primitive: null argument"); return RTHOST_ERR_INVALID; }
    const uint64_t n_new = n_indices / 3; // scene.rs:520: trailing partial triple is dropped
    if (s.tris.size() + n_new > 0xFFFFFFFFull) { set_error("add_primitive: triangle index overflows u32"); return RTHOST_ERR_LIMIT; }
    for (uint64_t i = 0; i < n_new * 3; i++)
        if (indices[i] >= n_verts) { set_error("add_primitive: vertex index out of range"); return RTHOST_ERR_INVALID; }

    std::vector<float> packed_normals;
    if (normals) { // scene.rs:506-511: normalise (times reciprocal length), then pack
        packed_normals.resize((size_t)n_verts * 2);
        for (uint32_t v = 0; v < n_verts; v++) {
            const float* n = normals + (size_t)v * 3;
            const float len = std::sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
            const float rl = normals_are_unit ? 1.0f : 1.0f / len;
            const float nn[3] = {n[0] * rl, n[1] * rl, n[2] * rl};
            pack_oct(nn, &packed_normals[(size_t)v * 2]);
        }
    }

    const uint32_t first_new = (uint32_t)s.tris.size();
    s.tris.resize(first_new + n_new);
    s.tri_exts.resize(first_new + n_new);
    for (uint64_t t = 0; t < n_new; t++) { // scene.rs:520-545
        const uint32_t idx[3] = {indices[t * 3], indices[t * 3 + 1], indices[t * 3 + 2]};
        TriExt ext;
        std::memset(&ext, 0, sizeof ext);
        for (int k = 0; k < 3; k++) {
            if (colors) ext.v[k].color = colors[idx[k]];
            if (normals) { ext.v[k].normal[0] = packed_normals[(size_t)idx[k] * 2]; ext.v[k].normal[1] = packed_normals[(size_t)idx[k] * 2 + 1]; }
            if (uv0) { ext.v[k].tc[0][0] = uv0[(size_t)idx[k] * 2]; ext.v[k].tc[0][1] = uv0[(size_t)idx[k] * 2 + 1]; }
            if (uv1) { ext.v[k].tc[1][0] = uv1[(size_t)idx[k] * 2]; ext.v[k].tc[1][1] = uv1[(size_t)idx[k] * 2 + 1]; }
        }
        tri_new(positions + (size_t)idx[0] * 3, positions + (size_t)idx[1] * 3, positions + (size_t)idx[2] * 3,
                s.tris[first_new + t]);
        s.tri_exts[first_new + t] = ext;
    }

    // scene.rs:553-564: build, flatten, append, rebase
    std::vector<Node>     nodes((size_t)2 * n_new + 1);
    std::vector<uint32_t> order(n_new ? n_new : 1);
    const uint32_t n_nodes = bvh_build(reinterpret_cast<const float*>(s.tris.data()), first_new, (uint32_t)n_new,
                                       nodes.data(), order.data(), s.build_threads);
    {
        std::vector<Tri>    tn(n_new);
        std::vector<TriExt> en(n_new);
        for (uint64_t i = 0; i < n_new; i++) { tn[i] = s.tris[order[i]]; en[i] = s.tri_exts[order[i]]; }
        std::copy(tn.begin(), tn.end(), s.tris.begin() + first_new);
        std::copy(en.begin(), en.end(), s.tri_exts.begin() + first_new);
    }
    s.tri_order.insert(s.tri_order.end(), order.begin(), order.begin() + n_new);
    if (s.nodes.size() + n_nodes > 0xFFFFFFFFull) { set_error("add_primitive: node index overflows u32"); return RTHOST_ERR_LIMIT; }
    const uint32_t bvh_root = (uint32_t)s.nodes.size();
    for (uint32_t k = 0; k < n_nodes; k++) {
        if (nodes[k].count == 0) nodes[k].first += bvh_root;
        s.nodes.push_back(nodes[k]);
    }

    Primitive p;
    std::memset(&p, 0, sizeof p);
    std::memcpy(p.transform, transform, sizeof p.transform);
    mat4_inverse_f64(transform, p.inv_transform); // Primitive::new, scene.rs:192-200
    p.material = material;
    p.bvh_idx = bvh_root;
    s.prims.push_back(p);
    if (out_prim) *out_prim = (uint32_t)s.prims.size() - 1;
    return RTHOST_OK;
}

int add_instance(Scene& s, uint32_t prim, const float transform[16], uint32_t* out_prim) {
    if (prim >= s.prims.size()) { set_error("add_instance: no such primitive"); return RTHOST_ERR_INVALID; }
    Primitive p = s.prims[prim]; // scene.rs:380-383
    std::memcpy(p.transform, transform, sizeof p.transform);
    mat4_inverse_f32(transform, p.inv_transform);
    s.prims.push_back(p);
    if (out_prim) *out_prim = (uint32_t)s.prims.size() - 1;
    return RTHOST_OK;
}

// ---------------------------------------------------------------- camera

static void camera_forward(const rthost_camera& c, float out[3]) {
    // Mat3::from_euler(ZYX, yaw, -pitch, roll) * FORWARD (input.rs:80-98): first column of
    // Rz(yaw) * Ry(-pitch) * Rx(roll)
    const float a = c.yaw, b = -c.pitch;
    out[0] = std::cos(a) * std::cos(b);
    out[1] = std::sin(a) * std::cos(b);
    out[2] = -std::sin(b);
}

// ---------------------------------------------------------------- CPU ray cast (autofocus)

namespace {
struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

bool tri_hit(const Tri& t, V3 ro, V3 rd, float& out) { // Tri::closest_hit, scene.rs:738-763
    const V3 p0{t.v[0][0], t.v[0][1], t.v[0][2]};
    const V3 e1 = sub(V3{t.v[1][0], t.v[1][1], t.v[1][2]}, p0), e2 = sub(V3{t.v[2][0], t.v[2][1], t.v[2][2]}, p0);
    const V3 h = cross(rd, e2);
    const float a = dot(e1, h);
    if (a > -0.000002f && a < 0.000002f) return false;
    const float f = 1.0f / a;
    const V3 sv = sub(ro, p0);
    const float u = f * dot(sv, h);
    if (u < 0.0f || u > 1.0f) return false;
    const V3 q = cross(sv, e1);
    const float v = f * dot(rd, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float tt = f * dot(e2, q);
    if (tt > 0.000002f) { out = tt; return true; }
    return false;
}

bool box_hit(const Node& n, V3 ro, V3 rd) { // Aabb::closest_hit, scene.rs:814-836
    const float* b = n.aabb;
    if (ro.x > b[0] && ro.y > b[1] && ro.z > b[2] && ro.x < b[3] && ro.y < b[4] && ro.z < b[5]) return true;
    const float r0x = (b[0] - ro.x) / rd.x, r0y = (b[1] - ro.y) / rd.y, r0z = (b[2] - ro.z) / rd.z;
    const float r1x = (b[3] - ro.x) / rd.x, r1y = (b[4] - ro.y) / rd.y, r1z = (b[5] - ro.z) / rd.z;
    const float t0 = std::fmax(std::fmin(r0x, r1x), std::fmax(std::fmin(r0y, r1y), std::fmin(r0z, r1z)));
    const float t1 = std::fmin(std::fmax(r0x, r1x), std::fmin(std::fmax(r0y, r1y), std::fmax(r0z, r1z)));
    return !(t0 >= t1 || t0 < 0.0f);
}
} // namespace

static bool scene_closest_hit(const Scene& s, const float ro_w[3], const float rd_w[3], float& out) {
    bool any = false;
    float closest = 0.0f;
    for (const Primitive& p : s.prims) { // scene.rs:640-658
        const float* m = p.inv_transform;
        const V3 ro{((m[0] * ro_w[0] + m[4] * ro_w[1]) + m[8] * ro_w[2]) + m[12],
                    ((m[1] * ro_w[0] + m[5] * ro_w[1]) + m[9] * ro_w[2]) + m[13],
                    ((m[2] * ro_w[0] + m[6] * ro_w[1]) + m[10] * ro_w[2]) + m[14]};
        const V3 d{(m[0] * rd_w[0] + m[4] * rd_w[1]) + m[8] * rd_w[2], (m[1] * rd_w[0] + m[5] * rd_w[1]) + m[9] * rd_w[2],
                   (m[2] * rd_w[0] + m[6] * rd_w[1]) + m[10] * rd_w[2]};
        const float len = std::sqrt(dot(d, d));
        const float rl = 1.0f / len;
        const V3 rd{d.x * rl, d.y * rl, d.z * rl};
        // Bvh::closest_hit_unindexed, scene.rs:1095-1136
        std::vector<uint32_t> stack{p.bvh_idx};
        float best = std::numeric_limits<float>::max();
        bool hit = false;
        while (!stack.empty()) {
            const Node& n = s.nodes[stack.back()];
            stack.pop_back();
            if (!box_hit(n, ro, rd)) continue;
            if (n.count > 0) {
                for (uint32_t i = 0; i < n.count; i++) {
                    float t;
                    if (tri_hit(s.tris[n.first + i], ro, rd, t) && t < best) { best = t; hit = true; }
                }
            } else {
                stack.push_back(n.first);
                stack.push_back(n.first + 1);
            }
        }
        if (hit) {
            const float tw = best / len;
            closest = any ? std::fmin(closest, tw) : std::fmin(std::numeric_limits<float>::max(), tw);
            any = true;
        }
    }
    out = closest;
    return any;
}

} // namespace rthost

// ==================================================================== C ABI

using namespace rthost;

// No exception may cross the C ABI: the loaders parse files nobody vouches for, and an allocation
// they ask for can fail (std::bad_alloc, std::length_error).  Every entry point that allocates runs
// its body through this.
template <class F> static int guarded(const char* what, F body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_error(std::string(what) + ": out of memory");
        return RTHOST_ERR_LIMIT;
    } catch (const std::exception& e) {
        set_error(std::string(what) + ": " + e.what());
        return RTHOST_ERR_FORMAT;
    } catch (...) {
        set_error(std::string(what) + ": unknown failure");
        return RTHOST_ERR_FORMAT;
    }
}

extern "C" {

const char* rthost_last_error(void) { return g_error.c_str(); }

int rthost_scene_new(rthost_scene** out) {
    return guarded("scene_new", [&]() -> int {
    if (!out) { set_error("scene_new: null out"); return RTHOST_ERR_INVALID; }
    *out = new rthost_scene();
    return RTHOST_OK;
    });
}
void rthost_scene_free(rthost_scene* s) { delete s; }

int rthost_scene_view_get(const rthost_scene* sc, rthost_scene_view* v) {
    if (!sc || !v) { set_error("scene_view_get: null argument"); return RTHOST_ERR_INVALID; }
    const Scene& s = sc->s;
    v->tris = s.tris.data();           v->n_tris = s.tris.size();
    v->tri_exts = s.tri_exts.data();
    v->nodes = s.nodes.data();         v->n_nodes = s.nodes.size();
    v->prims = s.prims.data();         v->n_prims = (uint32_t)s.prims.size();
    v->texels = s.texels.data();       v->n_texels = s.texels.size();
    v->env = s.env.data();             v->env_w = s.env_w; v->env_h = s.env_h;
    v->tri_order = s.tri_order.data();
    v->n_cameras = (uint32_t)s.cameras.size();
    return RTHOST_OK;
}

int rthost_scene_add_texture(rthost_scene* sc, const uint8_t* rgba8, uint32_t w, uint32_t h, uint32_t out_ref[2]) {
    return guarded("add_texture", [&]() -> int {
    if (!sc || !out_ref) { set_error("add_texture: null argument"); return RTHOST_ERR_INVALID; }
    TexRef r;
    int rc = add_texture(sc->s, rgba8, w, h, &r);
    if (rc == RTHOST_OK) { out_ref[0] = r.offset; out_ref[1] = r.size; }
    return rc;
    });
}

int rthost_scene_add_primitive(rthost_scene* sc, const float* positions, uint32_t n_verts, const uint32_t* indices,
                               uint64_t n_indices, const float* normals, const float* uv0, const float* uv1,
                               const uint32_t* colors, const void* material96, const float transform[16],
                               uint32_t* out_prim) {
    return guarded("add_primitive", [&]() -> int {
    if (!sc || !material96) { set_error("add_primitive: null argument"); return RTHOST_ERR_INVALID; }
    Material m;
    std::memcpy(&m, material96, sizeof m);
    return add_primitive(sc->s, positions, n_verts, indices, n_indices, normals, uv0, uv1, colors, m, transform, out_prim);
    });
}

int rthost_scene_add_instance(rthost_scene* sc, uint32_t prim, const float transform[16], uint32_t* out_prim) {
    return guarded("add_instance", [&]() -> int {
    if (!sc || !transform) { set_error("add_instance: null argument"); return RTHOST_ERR_INVALID; }
    return add_instance(sc->s, prim, transform, out_prim);
    });
}

int rthost_scene_set_env(rthost_scene* sc, const float* rgba32f, uint32_t w, uint32_t h) {
    return guarded("set_env", [&]() -> int {
    if (!sc || !rgba32f || !w || !h) { set_error("set_env: bad argument"); return RTHOST_ERR_INVALID; }
    sc->s.env.assign(rgba32f, rgba32f + (size_t)w * h * 4);
    sc->s.env_w = w;
    sc->s.env_h = h;
    return RTHOST_OK;
    });
}

int rthost_scene_load_gltf(rthost_scene* sc, const void* bytes, size_t n, const char* base_dir) {
    return guarded("load_gltf", [&]() -> int {
    if (!sc || !bytes) { set_error("load_gltf: null argument"); return RTHOST_ERR_INVALID; }
    return load_gltf(sc->s, static_cast<const uint8_t*>(bytes), n, base_dir ? base_dir : ".");
    });
}

int rthost_scene_load_gltf_file(rthost_scene* sc, const char* path) {
    return guarded("load_gltf_file", [&]() -> int {
    if (!sc || !path) { set_error("load_gltf_file: null argument"); return RTHOST_ERR_INVALID; }
    std::vector<uint8_t> bytes;
    if (!read_file(path, bytes)) { set_error(std::string("cannot read ") + path); return RTHOST_ERR_IO; }
    std::string p(path);
    size_t slash = p.find_last_of('/');
    return load_gltf(sc->s, bytes.data(), bytes.size(), slash == std::string::npos ? "." : p.substr(0, slash));
    });
}

int rthost_scene_load_hdr_file(rthost_scene* sc, const char* path) {
    return guarded("load_hdr_file", [&]() -> int {
    if (!sc || !path) { set_error("load_hdr_file: null argument"); return RTHOST_ERR_INVALID; }
    std::vector<uint8_t> bytes;
    if (!read_file(path, bytes)) { set_error(std::string("cannot read ") + path); return RTHOST_ERR_IO; }
    std::string err;
    if (!decode_hdr(bytes.data(), bytes.size(), sc->s.env, sc->s.env_w, sc->s.env_h, err)) {
        set_error("hdr: " + err);
        return RTHOST_ERR_FORMAT;
    }
    return RTHOST_OK;
    });
}

void rthost_camera_default(rthost_camera* c) { // input.rs:202-219
    const float deg = 0.017453292519943295f;
    c->pitch = -45.0f * deg;
    c->yaw = 0.0f;
    c->roll = 0.0f;
    c->position[0] = -5.0f; c->position[1] = 0.0f; c->position[2] = 5.0f;
    c->focus = 1.0f;
    c->fovy = 90.0f * deg;
    c->aperture = c->exposure = c->bloom = c->dispersion = 1.0f;
    c->aspect = 0.0f;
}

void rthost_camera_to_gpu(const rthost_camera* c, void* out48) { // input.rs:100-111
    GpuCamera g;
    camera_forward(*c, g.dir);
    g.fovy = c->fovy;
    std::memcpy(g.origin, c->position, sizeof g.origin);
    g.focus = c->focus;
    g.aperture = c->aperture; g.exposure = c->exposure; g.bloom = c->bloom; g.dispersion = c->dispersion;
    std::memcpy(out48, &g, sizeof g);
}

int rthost_scene_add_camera(rthost_scene* sc, const rthost_camera* cam) {
    return guarded("add_camera", [&]() -> int {
    if (!sc || !cam) { set_error("add_camera: null argument"); return RTHOST_ERR_INVALID; }
    sc->s.cameras.push_back(*cam);
    return RTHOST_OK;
    });
}

int rthost_scene_get_camera(const rthost_scene* sc, uint32_t idx, rthost_camera* out) {
    if (!sc || !out || idx >= sc->s.cameras.size()) { set_error("get_camera: bad argument"); return RTHOST_ERR_INVALID; }
    *out = sc->s.cameras[idx];
    return RTHOST_OK;
}

int rthost_scene_focus_camera(rthost_scene* sc, uint32_t idx, int* hit) { // scene.rs:670-677
    if (!sc || idx >= sc->s.cameras.size()) { set_error("focus_camera: bad argument"); return RTHOST_ERR_INVALID; }
    rthost_camera& c = sc->s.cameras[idx];
    float fwd[3], t;
    camera_forward(c, fwd);
    const bool h = scene_closest_hit(sc->s, c.position, fwd, t);
    if (h && c.focus != t) c.focus = std::max(t, 0.0f); // input.rs:143-148
    if (hit) *hit = h ? 1 : 0;
    return RTHOST_OK;
}

int rthost_frame_uniforms(const rthost_scene* sc, uint32_t camera_idx, const void* gpu_camera48, uint32_t res_x,
                          uint32_t res_y, uint32_t frame, float time, uint32_t reject_hist, void* out608) {
    if (!sc || !out608) { set_error("frame_uniforms: null argument"); return RTHOST_ERR_INVALID; }
    const Scene& s = sc->s;
    uint8_t* u = static_cast<uint8_t*>(out608);
    std::memset(u, 0, RTHOST_UNIFORMS_BYTES);
    if (gpu_camera48) {
        std::memcpy(u + 512, gpu_camera48, 48);
    } else {
        rthost_camera c;
        if (camera_idx < s.cameras.size()) c = s.cameras[camera_idx];
        else rthost_camera_default(&c); // scene.rs:698-702
        rthost_camera_to_gpu(&c, u + 512);
    }
    const uint32_t words[] = {(uint32_t)s.tris.size(), 0u, 0u, 0u, res_x, res_y, frame};
    std::memcpy(u + 560, words, sizeof words);
    std::memcpy(u + 588, &time, 4);
    const uint32_t tail[] = {reject_hist, (uint32_t)s.nodes.size(), (uint32_t)s.prims.size(), 0u};
    std::memcpy(u + 592, tail, sizeof tail);
    return RTHOST_OK;
}

int rthost_bvh_build(const void* tris48, uint32_t offset, uint32_t size, void* nodes_out, uint32_t* indices_out,
                     uint32_t* n_nodes_out, int n_threads) {
    return guarded("bvh_build", [&]() -> int {
    if (!tris48 || !nodes_out || !indices_out || !n_nodes_out) { set_error("bvh_build: null argument"); return RTHOST_ERR_INVALID; }
    if ((uint64_t)offset + size > 0xFFFFFFFFull) { set_error("bvh_build: range overflows u32"); return RTHOST_ERR_LIMIT; }
    *n_nodes_out = bvh_build(static_cast<const float*>(tris48), offset, size, static_cast<Node*>(nodes_out), indices_out, n_threads);
    return RTHOST_OK;
    });
}

void rthost_tri_new(const float p1[3], const float p2[3], const float p3[3], void* out48) {
    Tri t;
    tri_new(p1, p2, p3, t);
    std::memcpy(out48, &t, sizeof t);
}
void     rthost_pack_oct(const float n[3], float out[2]) { pack_oct(n, out); }
uint32_t rthost_pack_alpha(uint32_t mode, float cutoff) { return pack_alpha(mode, cutoff); }
void     rthost_from_gltf_mat4(const float g[16], float out[16]) { from_gltf_mat4(g, out); }
void     rthost_mat4_inverse_f64(const float m[16], float out[16]) { mat4_inverse_f64(m, out); }
void     rthost_mat4_inverse_f32(const float m[16], float out[16]) { mat4_inverse_f32(m, out); }

} // extern "C"
