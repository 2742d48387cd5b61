/*
 * rt_oracle.c -- CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 * See rt_oracle.h for scope and the "parity unpinned" statement.
 *
 * Build with:  gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * Every float operation is IEEE binary32, round-to-nearest, never contracted into an FMA and
 * never reassociated: that is the Rust semantics of the host code being restated (scene.rs)
 * and the evaluation order chosen for the WGSL expressions (left to right, as written).
 * Transcendentals come from the C library (sinf, cosf, atan2f, asinf, powf).
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

_Static_assert(sizeof(OrcTri) == 48, "Tri");
_Static_assert(sizeof(OrcTriExt) == 96, "TriExt");
_Static_assert(sizeof(OrcNode) == 32, "BvhNode");
_Static_assert(sizeof(OrcMaterial) == 96, "Material");
_Static_assert(sizeof(OrcPrimitive) == 240, "Primitive");
_Static_assert(sizeof(OrcCamera) == 48, "GpuCamera");
_Static_assert(sizeof(OrcFrameUniforms) == 608, "FrameUniforms");

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { return vdivs(a, vlength(a)); }
static inline v3 vrcp(v3 a) { return V(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float fractf(float x) { return x - floorf(x); }
static inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
/* WGSL f32 -> u32 conversion: clamp to the target range, truncate; NaN -> 0 */
static inline uint32_t f2u(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

/* ------------------------------------------------------------------ RNG */

/* shader.wgsl:641-647 */
uint32_t orc_hash21(uint32_t x, uint32_t y) {
    x *= 73333u;
    y *= 7777u;
    x ^= 3333777777u >> (x >> 28);
    y ^= 3333777777u >> (y >> 28);
    uint32_t n = x * y;
    return n ^ (n >> 15);
}

/* shader.wgsl:650-659 */
uint32_t orc_rand_u32(uint32_t* seed) {
    uint32_t old = *seed;
    *seed = orc_hash21(old, old ^ 39213742u);
    return orc_hash21(old, *seed);
}

float orc_rand(uint32_t* seed) {
    /* 1.0 / f32(0xffffffffu): the u32 rounds up to 2^32, so the scale is exactly 2^-32 */
    return (float)orc_rand_u32(seed) * (1.0f / 4294967296.0f);
}

/* shader.wgsl:895 */
uint32_t orc_pixel_seed(uint32_t px, uint32_t py, uint32_t frame) {
    return orc_hash21(orc_hash21(px, py), frame);
}

/* ------------------------------------------------------------------ BVH builder */

typedef struct { float d[6]; } box6;

/* scene.rs:800-804 */
static box6 box_point(const float* p) {
    box6 b = {{p[0] - 0.00001f, p[1] - 0.00001f, p[2] - 0.00001f,
               p[0] + 0.00001f, p[1] + 0.00001f, p[2] + 0.00001f}};
    return b;
}
/* scene.rs:785-794 */
static box6 box_with(box6 a, box6 b) {
    box6 r;
    for (int i = 0; i < 3; i++) r.d[i] = fminf(a.d[i], b.d[i]);
    for (int i = 3; i < 6; i++) r.d[i] = fmaxf(a.d[i], b.d[i]);
    return r;
}
/* scene.rs:727-731 */
static box6 tri_box(const OrcTri* t) {
    return box_with(box_with(box_point(t->v[0]), box_point(t->v[1])), box_point(t->v[2]));
}
/* scene.rs:780-783 */
static float box_surface(const box6* b) {
    float sx = b->d[3] - b->d[0], sy = b->d[4] - b->d[1], sz = b->d[5] - b->d[2];
    return (sx * sy + sy * sz + sz * sx) * 2.0f;
}

/* scene.rs:720-725 */
void orc_tri_new(const float p1[3], const float p2[3], const float p3[3], OrcTri* out) {
    for (int k = 0; k < 3; k++) {
        float c = ((p1[k] + p2[k]) + p3[k]) / 3.0f;
        out->v[0][k] = p1[k];
        out->v[1][k] = p2[k];
        out->v[2][k] = p3[k];
        out->v[k][3] = c;
    }
}

typedef struct {
    const OrcTri* tris;
    OrcNode*      nodes;
    uint32_t      n_nodes;
    uint32_t*     indices; /* indices[i - offset] = global triangle id */
    size_t        offset;
} Builder;

static inline float tri_centroid(const OrcTri* t, int axis) { return t->v[axis][3]; }

/* scene.rs:870-877 */
static void node_update_aabb(const Builder* b, OrcNode* n) {
    if (n->count == 0) return;
    box6 bb = tri_box(&b->tris[b->indices[n->first - b->offset]]);
    for (uint32_t i = n->first; i < n->first + n->count; i++)
        bb = box_with(bb, tri_box(&b->tris[b->indices[i - b->offset]]));
    memcpy(n->aabb, bb.d, sizeof bb.d);
}

/* scene.rs:921-946 */
static float evaluate_split(const Builder* b, const OrcNode* node, int axis, float split) {
    /* Aabb::new, scene.rs:774-778: Rust f32::MIN is the most negative finite float */
    box6 lb = {{FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}};
    box6 rb = lb;
    float lc = 0.0f, rc = 0.0f; /* f32 counters, incremented by 1.0 (saturate at 2^24) */
    for (uint32_t i = node->first; i < node->first + node->count; i++) {
        const OrcTri* t = &b->tris[b->indices[i - b->offset]];
        if (tri_centroid(t, axis) < split) {
            lc += 1.0f;
            lb = box_with(lb, tri_box(t));
        } else {
            rc += 1.0f;
            rb = box_with(rb, tri_box(t));
        }
    }
    float cost = lc * box_surface(&lb) + rc * box_surface(&rb);
    return cost > 0.0f ? cost : FLT_MAX; /* NaN (empty side: 0*inf) fails the test */
}

/* scene.rs:948-967 */
static void find_best_split(const Builder* b, const OrcNode* node, int* axis_out, float* split_out) {
    int best_axis = 0;
    float best_split = 0.0f, best_cost = FLT_MAX;
    for (int axis = 0; axis < 3; axis++) {
        for (uint32_t i = node->first; i < node->first + node->count; i++) {
            float split = tri_centroid(&b->tris[b->indices[i - b->offset]], axis);
            float cost = evaluate_split(b, node, axis, split);
            if (cost < best_cost) { best_axis = axis; best_cost = cost; best_split = split; }
        }
    }
    *axis_out = best_axis;
    *split_out = best_split;
}

/* scene.rs:969-987 */
static void find_split_approx(const Builder* b, const OrcNode* node, int count, int* axis_out,
                              float* split_out) {
    int best_axis = 0;
    float best_split = 0.0f, best_cost = FLT_MAX;
    for (int axis = 0; axis < 3; axis++) {
        for (int i = 0; i < count; i++) {
            float mn = node->aabb[axis], mx = node->aabb[3 + axis];
            float split = mn + (((float)i + 0.5f) / (float)count) * (mx - mn);
            float cost = evaluate_split(b, node, axis, split);
            if (cost < best_cost) { best_axis = axis; best_cost = cost; best_split = split; }
        }
    }
    *axis_out = best_axis;
    *split_out = best_split;
}

/* scene.rs:991-1050 */
static void subdivide(Builder* b, uint32_t node_idx) {
    OrcNode node = b->nodes[node_idx];
    if (node.count <= 2) return;

    int axis;
    float split;
    if (node.count < 64) find_best_split(b, &node, &axis, &split);
    else find_split_approx(b, &node, 16, &axis, &split);

    size_t i = node.first;
    size_t j = (size_t)node.first + node.count - 1;
    while (i <= j) {
        uint32_t first_idx = b->indices[i - b->offset];
        if (tri_centroid(&b->tris[first_idx], axis) < split) {
            i += 1;
        } else {
            uint32_t tmp = b->indices[i - b->offset];
            b->indices[i - b->offset] = b->indices[j - b->offset];
            b->indices[j - b->offset] = tmp;
            if (j == 0) break;
            j -= 1;
        }
    }

    /* BvhNode::new, scene.rs:854-860 */
    float m100[3] = {-100.0f, -100.0f, -100.0f};
    box6 init = box_point(m100);
    OrcNode left;
    memcpy(left.aabb, init.d, sizeof init.d);
    left.first = node.first;
    left.count = (uint32_t)i - node.first;
    node_update_aabb(b, &left);
    if (left.count == 0 || left.count == node.count) return;

    OrcNode right;
    memcpy(right.aabb, init.d, sizeof init.d);
    right.first = (uint32_t)i;
    right.count = node.count - left.count;
    node_update_aabb(b, &right);

    uint32_t children = b->n_nodes;
    b->nodes[node_idx].count = 0;
    b->nodes[node_idx].first = children;
    b->nodes[b->n_nodes++] = left;
    b->nodes[b->n_nodes++] = right;

    subdivide(b, children);
    subdivide(b, children + 1);
}

/* scene.rs:888-899 */
uint32_t orc_bvh_build(const OrcTri* tris, uint32_t offset, uint32_t size, OrcNode* nodes_out,
                       uint32_t* indices_out) {
    Builder b;
    b.tris = tris;
    b.nodes = nodes_out;
    b.n_nodes = 0;
    b.indices = indices_out;
    b.offset = offset;
    for (uint32_t i = 0; i < size; i++) indices_out[i] = offset + i;

    OrcNode root;
    float m100[3] = {-100.0f, -100.0f, -100.0f};
    box6 init = box_point(m100);
    memcpy(root.aabb, init.d, sizeof init.d);
    root.first = offset;
    root.count = size;
    node_update_aabb(&b, &root);
    b.nodes[b.n_nodes++] = root;
    subdivide(&b, 0);
    return b.n_nodes;
}

/* scene.rs:553-564 with flatten_triangles (scene.rs:902-919) */
uint32_t orc_build_primitive(OrcTri* tris, OrcTriExt* exts, uint32_t offset, uint32_t size,
                             OrcNode* nodes_out, uint32_t bvh_root, uint32_t* order_out) {
    uint32_t* indices = (uint32_t*)malloc(sizeof(uint32_t) * (size ? size : 1));
    uint32_t n = orc_bvh_build(tris, offset, size, nodes_out, indices);
    OrcTri* tn = (OrcTri*)malloc(sizeof(OrcTri) * (size ? size : 1));
    OrcTriExt* en = (OrcTriExt*)malloc(sizeof(OrcTriExt) * (size ? size : 1));
    for (uint32_t i = 0; i < size; i++) {
        tn[i] = tris[indices[i]];
        en[i] = exts[indices[i]];
        if (order_out) order_out[i] = indices[i] - offset;
    }
    memcpy(tris + offset, tn, sizeof(OrcTri) * size);
    memcpy(exts + offset, en, sizeof(OrcTriExt) * size);
    for (uint32_t k = 0; k < n; k++)
        if (nodes_out[k].count == 0) nodes_out[k].first += bvh_root;
    free(indices);
    free(tn);
    free(en);
    return n;
}

/* ------------------------------------------------------------------ host packing helpers */

static inline float signumf(float x) { /* Rust f32::signum: 1.0 for +0, -1.0 for -0 */
    if (x != x) return x;
    return signbit(x) ? -1.0f : 1.0f;
}

/* scene.rs:203-221 */
void orc_pack_oct(const float n_in[3], float out[2]) {
    float s = (fabsf(n_in[0]) + fabsf(n_in[1])) + fabsf(n_in[2]);
    float nx = n_in[0] / s, ny = n_in[1] / s, nz = n_in[2] / s;
    if (nz < 0.0f) {
        float wx = (1.0f - fabsf(ny)) * signumf(nx);
        float wy = (1.0f - fabsf(nx)) * signumf(ny);
        nx = wx;
        ny = wy;
    }
    out[0] = nx * 0.5f + 0.5f;
    out[1] = ny * 0.5f + 0.5f;
}

/* scene.rs:170-177 */
uint32_t orc_pack_alpha(uint32_t mode, float cutoff) {
    float c = cutoff < 0.0f ? 0.0f : (cutoff > 1.0f ? 1.0f : cutoff);
    return mode | ((uint32_t)(c * 65535.0f) << 16);
}

/* scene.rs:195: transform.as_dmat4().inverse().as_mat4() -- cofactor expansion in f64 */
void orc_mat4_inverse_f64(const float mf[16], float out[16]) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = (double)mf[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double rdet = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * rdet);
}

/* input.rs:96-111: dir = Mat3::from_euler(ZYX, yaw, -pitch, roll) * (1,0,0) */
void orc_camera_dir(float yaw, float pitch, float out[3]) {
    out[0] = cosf(yaw) * cosf(pitch);
    out[1] = sinf(yaw) * cosf(pitch);
    out[2] = sinf(pitch);
}

/* input.rs:202-219 */
void orc_camera_default(OrcCamera* c) {
    const float deg = 0.017453292519943295f;
    orc_camera_dir(0.0f, -45.0f * deg, c->dir);
    c->fovy = 90.0f * deg;
    c->origin[0] = -5.0f;
    c->origin[1] = 0.0f;
    c->origin[2] = 5.0f;
    c->focus = 1.0f;
    c->aperture = c->exposure = c->bloom = c->dispersion = 1.0f;
}

/* ------------------------------------------------------------------ the path */

typedef struct { v3 origin, dir, idir; } Ray;

typedef struct {
    float       t;
    int32_t     idx;
    int32_t     prim_idx;
    v3          normal;
    v3          bary;
} Hit;

typedef struct {
    const OrcScene*  s;
    const OrcParams* p;
    uint32_t         seed;
    float            debug; /* shader.wgsl:440 */
    OrcCounters      c;
} Ctx;

static inline float rnd(Ctx* cx) {
    cx->c.rng_draws++;
    return orc_rand(&cx->seed);
}

static const float PI_F = 3.141592654f; /* shader.wgsl:11 */

/* shader.wgsl:366-372 */
static inline float sign11(float x) { return x < 0.0f ? -1.0f : 1.0f; }

/* shader.wgsl:175-182 */
static void unpack_rgba8(uint32_t x, float out[4]) {
    out[0] = (float)((x >> 24) & 255u) / 255.0f;
    out[1] = (float)((x >> 16) & 255u) / 255.0f;
    out[2] = (float)((x >> 8) & 255u) / 255.0f;
    out[3] = (float)(x & 255u) / 255.0f;
}

/* shader.wgsl:144-160.  size == 0 returns the placeholder (:215-226) whose colour is
 * never consumed by the path; only its alpha (1.0) is.  The flat texel index follows the
 * reference arithmetic (so fract() == 1.0 runs into the next row) and is then clamped to
 * the array, the defined choice for the reference's out-of-bounds read. */
static void sample_texture(Ctx* cx, OrcTexRef tex, float tcx, float tcy, float out[4]) {
    if (tex.size == 0) {
        out[0] = 0.8f; out[1] = 0.3f; out[2] = 0.3f; out[3] = 1.0f;
        return;
    }
    uint32_t w = tex.size >> 16, h = tex.size & 0xFFFFu;
    uint32_t px = f2u(fractf(tcx) * (float)w);
    uint32_t py = f2u(fractf(tcy) * (float)h);
    uint64_t idx = (uint64_t)(uint32_t)(tex.offset + py * w + px);
    if (idx >= cx->s->n_texels) idx = cx->s->n_texels ? cx->s->n_texels - 1 : 0;
    cx->c.texel_fetches++;
    unpack_rgba8(cx->s->n_texels ? cx->s->texels[idx] : 0u, out);
}

/* shader.wgsl:185-197 */
static v3 unpack_oct(const float f_in[2]) {
    float fx = f_in[0] * 2.0f - 1.0f, fy = f_in[1] * 2.0f - 1.0f;
    v3 n = V(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    float t = saturatef(-n.z);
    n.x -= t * sign11(n.x);
    n.y -= t * sign11(n.y);
    return vnormalize(n);
}

/* shader.wgsl:339-364 */
static float intersect(const Ray* ray, const OrcTri* tri) {
    v3 p0 = V(tri->v[0][0], tri->v[0][1], tri->v[0][2]);
    v3 e1 = vsub(V(tri->v[1][0], tri->v[1][1], tri->v[1][2]), p0);
    v3 e2 = vsub(V(tri->v[2][0], tri->v[2][1], tri->v[2][2]), p0);
    v3 h = vcross(ray->dir, e2);
    float a = vdot(e1, h);
    if (a > -0.000002f && a < 0.000002f) return -1.0f;
    float f = 1.0f / a;
    v3 s = vsub(ray->origin, p0);
    float u = f * vdot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    v3 q = vcross(s, e1);
    float v = f * vdot(ray->dir, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * vdot(e2, q);
    return t > 0.000002f ? t : -1.0f;
}

/* shader.wgsl:376-412 */
static Hit intersect_full(const Ctx* cx, const Ray* ray, int32_t idx) {
    const OrcTri* tri = &cx->s->tris[idx];
    Hit hit;
    hit.t = 0.0f; hit.idx = -1; hit.prim_idx = -1;
    hit.normal = V(0.0f, 0.0f, 1.0f);
    hit.bary = V(0.333f, 0.333f, 0.333f);
    v3 p0 = V(tri->v[0][0], tri->v[0][1], tri->v[0][2]);
    v3 e1 = vsub(V(tri->v[1][0], tri->v[1][1], tri->v[1][2]), p0);
    v3 e2 = vsub(V(tri->v[2][0], tri->v[2][1], tri->v[2][2]), p0);
    hit.normal = vnormalize(vcross(e1, e2));
    hit.normal = vscale(hit.normal, -sign11(vdot(hit.normal, ray->dir)));
    v3 h = vcross(ray->dir, e2);
    float a = vdot(e1, h);
    if (a > -0.000002f && a < 0.000002f) return hit;
    float f = 1.0f / a;
    v3 s = vsub(ray->origin, p0);
    float u = f * vdot(s, h);
    if (u < 0.0f || u > 1.0f) return hit;
    v3 q = vcross(s, e1);
    float v = f * vdot(ray->dir, q);
    if (v < 0.0f || u + v > 1.0f) return hit;
    float t = f * vdot(e2, q);
    if (t <= 0.000002f) return hit;
    hit.idx = idx;
    hit.t = t;
    hit.bary = V((1.0f - u) - v, u, v);
    return hit;
}

/* shader.wgsl:415-438 */
static float intersect_aabb(const Ray* ray, const float* bb) {
    v3 bmin = V(bb[0], bb[1], bb[2]), bmax = V(bb[3], bb[4], bb[5]);
    v3 o = ray->origin;
    if (o.x >= bmin.x && o.y >= bmin.y && o.z >= bmin.z && o.x < bmax.x && o.y < bmax.y &&
        o.z < bmax.z)
        return 0.0f;
    v3 rmin = vmul(vsub(bmin, o), ray->idir);
    v3 rmax = vmul(vsub(bmax, o), ray->idir);
    v3 tmin = V(fminf(rmin.x, rmax.x), fminf(rmin.y, rmax.y), fminf(rmin.z, rmax.z));
    v3 tmax = V(fmaxf(rmin.x, rmax.x), fmaxf(rmin.y, rmax.y), fmaxf(rmin.z, rmax.z));
    float t0 = fmaxf(tmin.x, fmaxf(tmin.y, tmin.z));
    float t1 = fminf(tmax.x, fminf(tmax.y, tmax.z));
    if (t0 > t1 || t0 < 0.0f) return -1.0f;
    return t0;
}

#define ORC_MAX_STACK 256

/* shader.wgsl:442-525 */
static int32_t trace_bvh(Ctx* cx, const Ray* ray, uint32_t root, float* t_max,
                         const OrcPrimitive* prim) {
    const OrcScene* s = cx->s;
    uint32_t stack[ORC_MAX_STACK];
    uint32_t sp = 0;
    uint32_t depth = cx->p->stack_depth < ORC_MAX_STACK ? cx->p->stack_depth : ORC_MAX_STACK;
    OrcNode node = s->nodes[root];
    float best_t = *t_max;
    int32_t best_i = -1;
    cx->c.root_tests++;
    if (intersect_aabb(ray, node.aabb) < -0.5f) return best_i;

    const uint32_t alpha_settings = prim->material.alpha_settings;
    for (;;) {
        cx->debug += 1.0f;
        if (node.count > 0) {
            cx->c.leaf_steps++;
            for (uint32_t i = node.first; i < node.first + node.count; i++) {
                cx->c.tri_tests++;
                float t = intersect(ray, &s->tris[i]);
                if (t >= 0.0f && t < best_t) {
                    if ((alpha_settings & 3u) != 0u) {
                        cx->c.alpha_tests++;
                        Hit hit = intersect_full(cx, ray, (int32_t)i);
                        const OrcTriExt* ext = &s->exts[i];
                        uint32_t set = prim->material.albedo_texcoord;
                        if (set > 1u) set = 1u; /* array<vec2f,2> index, clamped */
                        float tcx = 0.0f, tcy = 0.0f;
                        tcx += hit.bary.x * ext->v[0].tc[set][0];
                        tcy += hit.bary.x * ext->v[0].tc[set][1];
                        tcx += hit.bary.y * ext->v[1].tc[set][0];
                        tcy += hit.bary.y * ext->v[1].tc[set][1];
                        tcx += hit.bary.z * ext->v[2].tc[set][0];
                        tcy += hit.bary.z * ext->v[2].tc[set][1];
                        float texel[4];
                        sample_texture(cx, prim->material.albedo, tcx, tcy, texel);
                        float alpha = texel[3];
                        if ((alpha_settings & 3u) == 1u) {
                            if (alpha < (float)(alpha_settings >> 16) / 65536.0f) continue;
                        } else {
                            if (rnd(cx) > alpha) continue;
                        }
                    }
                    best_i = (int32_t)i;
                    best_t = t;
                }
            }
            if (sp == 0) break;
            node = s->nodes[stack[--sp]];
        } else {
            cx->c.inner_steps++;
            float left = intersect_aabb(ray, s->nodes[node.first].aabb);
            float right = intersect_aabb(ray, s->nodes[node.first + 1].aabb);
            int lmiss = (left < -0.5f || left > best_t);
            int rmiss = (right < -0.5f || right > best_t);
            if (lmiss && rmiss) {
                if (sp == 0) break;
                node = s->nodes[stack[--sp]];
            } else if (lmiss) {
                node = s->nodes[node.first + 1];
            } else if (rmiss) {
                node = s->nodes[node.first];
            } else {
                uint32_t far_child = left < right ? node.first + 1 : node.first;
                uint32_t near_child = left < right ? node.first : node.first + 1;
                if (sp < depth) {
                    stack[sp++] = far_child;
                } else { /* shader.wgsl:306-313: push dropped, pixel flagged */
                    cx->debug = -99999999.0f;
                    cx->c.stack_overflows++;
                }
                node = s->nodes[near_child];
            }
        }
    }
    *t_max = best_t;
    return best_i;
}

/* shader.wgsl:577-595; mat4 * vec4 evaluated column by column, left to right */
static v3 transform_dir(v3 x, const float* m) {
    return V((m[0] * x.x + m[4] * x.y) + m[8] * x.z, (m[1] * x.x + m[5] * x.y) + m[9] * x.z,
             (m[2] * x.x + m[6] * x.y) + m[10] * x.z);
}
static v3 transform_pos(v3 x, const float* m) {
    return V(((m[0] * x.x + m[4] * x.y) + m[8] * x.z) + m[12],
             ((m[1] * x.x + m[5] * x.y) + m[9] * x.z) + m[13],
             ((m[2] * x.x + m[6] * x.y) + m[10] * x.z) + m[14]);
}
/* normalize(transpose(t_inv) * vec4(x, 0)) */
static v3 transform_normal(v3 x, const float* m) {
    return vnormalize(V((m[0] * x.x + m[1] * x.y) + m[2] * x.z, (m[4] * x.x + m[5] * x.y) + m[6] * x.z,
                        (m[8] * x.x + m[9] * x.y) + m[10] * x.z));
}
static Ray transform_ray(const Ray* x, const float* it) {
    Ray r;
    r.dir = vnormalize(transform_dir(x->dir, it));
    r.origin = transform_pos(x->origin, it);
    r.idir = vrcp(r.dir);
    return r;
}

/* shader.wgsl:598-627 */
static Hit trace(Ctx* cx, const Ray* ray) {
    const OrcScene* s = cx->s;
    int32_t closest_idx = -1;
    float closest_t = 99999999.0f;
    uint32_t closest_primitive = 0;
    cx->debug = 0.0f;
    cx->c.rays++;
    for (uint32_t i = 0; i < s->n_prims; i++) {
        const OrcPrimitive* pr = &s->prims[i];
        float scale = vlength(transform_dir(ray->dir, pr->inv_transform));
        Ray t_ray = transform_ray(ray, pr->inv_transform);
        float new_t = closest_t * scale;
        int32_t new_idx = trace_bvh(cx, &t_ray, pr->bvh_idx, &new_t, pr);
        if (new_idx >= 0) {
            closest_t = new_t / scale;
            closest_idx = new_idx;
            closest_primitive = i;
        }
    }
    Hit hit;
    if (closest_idx < 0) { /* the reference's triangles[-1] read cannot produce a hit (:408) */
        hit.t = closest_t; hit.idx = -1; hit.prim_idx = 0;
        hit.normal = V(0.0f, 0.0f, 1.0f);
        hit.bary = V(0.333f, 0.333f, 0.333f);
        return hit;
    }
    const OrcPrimitive* pr = &s->prims[closest_primitive];
    Ray t_final = transform_ray(ray, pr->inv_transform);
    hit = intersect_full(cx, &t_final, closest_idx);
    hit.t = closest_t;
    hit.normal = transform_normal(hit.normal, pr->inv_transform);
    hit.prim_idx = (int32_t)closest_primitive;
    return hit;
}

/* shader.wgsl:528-575, 630-638 (reached only by the opt-in sun extension) */
static int trace_shadow(Ctx* cx, const Ray* ray) {
    const OrcScene* s = cx->s;
    uint32_t depth = cx->p->stack_depth < ORC_MAX_STACK ? cx->p->stack_depth : ORC_MAX_STACK;
    cx->c.shadow_rays++;
    for (uint32_t p = 0; p < s->n_prims; p++) {
        Ray r = transform_ray(ray, s->prims[p].inv_transform);
        uint32_t stack[ORC_MAX_STACK];
        uint32_t sp = 0;
        OrcNode node = s->nodes[s->prims[p].bvh_idx];
        if (intersect_aabb(&r, node.aabb) < -0.5f) continue;
        for (;;) {
            if (node.count > 0) {
                for (uint32_t i = node.first; i < node.first + node.count; i++)
                    if (intersect(&r, &s->tris[i]) >= 0.0f) return 1;
                if (sp == 0) break;
                node = s->nodes[stack[--sp]];
            } else {
                float left = intersect_aabb(&r, s->nodes[node.first].aabb);
                float right = intersect_aabb(&r, s->nodes[node.first + 1].aabb);
                if (left < -0.5f && right < -0.5f) {
                    if (sp == 0) break;
                    node = s->nodes[stack[--sp]];
                } else if (left < -0.5f) {
                    node = s->nodes[node.first + 1];
                } else if (right < -0.5f) {
                    node = s->nodes[node.first];
                } else {
                    uint32_t far_child = left < right ? node.first + 1 : node.first;
                    uint32_t near_child = left < right ? node.first : node.first + 1;
                    if (sp < depth) stack[sp++] = far_child;
                    else cx->debug = -99999999.0f; /* push() -> assert(false), shader.wgsl:306-313: the pixel turns yellow (:941-943) */
                    node = s->nodes[near_child];
                }
            }
        }
    }
    return 0;
}

/* shader.wgsl:662-667 */
static v3 rand_sphere(Ctx* cx) {
    float theta = rnd(cx) * 2.0f * PI_F;
    float z = rnd(cx) * 2.0f - 1.0f;
    float radius = sqrtf(1.0f - z * z);
    return V(radius * cosf(theta), radius * sinf(theta), z);
}
/* shader.wgsl:675-681 */
static v3 rand_hemisphere(Ctx* cx, v3 normal) {
    v3 sp = rand_sphere(cx);
    if (vdot(sp, normal) < 0.0f) sp = vscale(sp, -1.0f);
    return sp;
}

/* shader.wgsl:691-702 */
static v3 sky(const Ctx* cx, v3 dir) {
    const OrcScene* s = cx->s;
    float u = atan2f(dir.y, dir.x) / (PI_F * 2.0f) + 0.5f;
    float v = asinf(-dir.z) / PI_F + 0.5f;
    uint32_t x = f2u(u * (float)s->env_w), y = f2u(v * (float)s->env_h);
    if (x >= s->env_w) x = s->env_w - 1; /* textureLoad out of range: coordinates clamped */
    if (y >= s->env_h) y = s->env_h - 1;
    const float* px = &s->env[((uint64_t)y * s->env_w + x) * 4];
    return V(px[0] * 1.0f, px[1] * 1.0f, px[2] * 1.0f);
}

/* shader.wgsl:741-779 */
static Ray camera_ray(Ctx* cx, const OrcFrameUniforms* u, uint32_t px, uint32_t py) {
    Ray ray;
    const OrcCamera* cam = &u->camera;
    ray.origin = V(cam->origin[0], cam->origin[1], cam->origin[2]);
    v3 forward = V(cam->dir[0], cam->dir[1], cam->dir[2]);
    float fov_factor = (sinf(cam->fovy / 2.0f) / cosf(cam->fovy / 2.0f)) * 2.0f;
    v3 right = vnormalize(vcross(forward, V(0.0f, 0.0f, 1.0f)));
    v3 up = vnormalize(vcross(right, forward));
    v3 pixel_pos = vadd(ray.origin, forward);
    float rx = rnd(cx);
    float ry = rnd(cx);
    float aax = (float)px + rx, aay = (float)py + ry;
    float resx = (float)u->res[0], resy = (float)u->res[1];
    float aspect = resx / resy;
    pixel_pos = vadd(pixel_pos, vscale(vscale(vscale(right, aax / resx - 0.5f), fov_factor), aspect));
    pixel_pos = vadd(pixel_pos, vscale(vscale(up, 0.5f - aay / resy), fov_factor));
    ray.dir = vnormalize(vsub(pixel_pos, ray.origin));
    /* rand_disk, shader.wgsl:669-673 */
    float theta = rnd(cx) * 2.0f * PI_F;
    float radius = sqrtf(rnd(cx));
    float ax = cx->p->aperture_radius * (radius * cosf(theta));
    float ay = cx->p->aperture_radius * (radius * sinf(theta));
    ray.origin = vadd(ray.origin, vscale(right, ax));
    ray.origin = vadd(ray.origin, vscale(up, ay));
    pixel_pos = vadd(pixel_pos, vscale(ray.dir, cam->focus - 1.0f));
    ray.dir = vnormalize(vsub(pixel_pos, ray.origin));
    ray.idir = vrcp(ray.dir);
    return ray;
}

static inline v3 to_linear(const float* c) { /* shader.wgsl:725-728 */
    return V(powf(c[0], 2.2f), powf(c[1], 2.2f), powf(c[2], 2.2f));
}

/* shader.wgsl:803-881 */
static void surface_hit_brdf(Ctx* cx, Ray* ray, const Hit* hit, v3* throughput, v3* lighting) {
    const OrcScene* s = cx->s;
    const OrcPrimitive* pr = &s->prims[hit->prim_idx];
    const OrcMaterial* m = &pr->material;
    cx->c.shade_calls++;
    (void)rnd(cx);

    ray->origin = vadd(ray->origin, vscale(ray->dir, hit->t));
    ray->origin = vadd(ray->origin, vscale(hit->normal, 0.001f));

    /* tri_ext_interpolate, shader.wgsl:229-253 */
    const OrcTriExt* ext = &s->exts[hit->idx];
    float b[3] = {hit->bary.x, hit->bary.y, hit->bary.z};
    float tc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    v3 sn = V(0.0f, 0.0f, 0.0f);
    for (int k = 0; k < 3; k++) {
        tc[0][0] += b[k] * ext->v[k].tc[0][0];
        tc[0][1] += b[k] * ext->v[k].tc[0][1];
        tc[1][0] += b[k] * ext->v[k].tc[1][0];
        tc[1][1] += b[k] * ext->v[k].tc[1][1];
        sn = vadd(sn, vscale(unpack_oct(ext->v[k].normal), b[k]));
    }
    sn = vnormalize(sn);

    if (vlength(sn) > 0.001f) {
        sn = transform_normal(vnormalize(sn), pr->inv_transform);
        if (vdot(sn, hit->normal) < 0.0f) sn = vneg(sn);
    } else {
        sn = hit->normal;
    }

    float texel[4];
    v3 albedo = V(m->albedo_factor[0], m->albedo_factor[1], m->albedo_factor[2]);
    if (m->albedo.size != 0) {
        uint32_t set = m->albedo_texcoord > 1u ? 1u : m->albedo_texcoord;
        sample_texture(cx, m->albedo, tc[set][0], tc[set][1], texel);
        albedo = vmul(albedo, to_linear(texel));
    } else {
        albedo = vmul(albedo, V(0.01f, 0.02f, 0.03f));
    }
    v3 emissive = V(m->emissive_factor[0], m->emissive_factor[1], m->emissive_factor[2]);
    if (m->emissive.size != 0) {
        uint32_t set = m->emissive_texcoord > 1u ? 1u : m->emissive_texcoord;
        sample_texture(cx, m->emissive, tc[set][0], tc[set][1], texel);
        emissive = vmul(emissive, to_linear(texel));
    }
    float metallic_chance = m->metallic_factor;
    float roughness = m->roughness_factor;
    if (m->metallic_roughness.size != 0) {
        uint32_t set = m->metal_r_texcoord > 1u ? 1u : m->metal_r_texcoord;
        sample_texture(cx, m->metallic_roughness, tc[set][0], tc[set][1], texel);
        metallic_chance = texel[2];
        roughness *= texel[1];
    }

    int metal = rnd(cx) < metallic_chance;
    v3 f0 = metal ? albedo : V(0.04f, 0.04f, 0.04f);

    v3 hemi = rand_hemisphere(cx, sn);
    float mt = roughness * roughness * 0.8f;
    v3 mn = vnormalize(V(mixf(sn.x, hemi.x, mt), mixf(sn.y, hemi.y, mt), mixf(sn.z, hemi.z, mt)));

    /* fresnel_schlick, shader.wgsl:791-793; only .x is consumed */
    float pw = powf(1.0f - vdot(mn, vneg(ray->dir)), 5.0f);
    float fresnel_x = f0.x + (1.0f - f0.x) * pw;
    int specular = rnd(cx) < fresnel_x;

    *lighting = vadd(*lighting, vmul(emissive, *throughput));
    if (!specular || metal) *throughput = vmul(*throughput, albedo);

    if (specular || metal) {
        /* reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2 */
        v3 d = vsub(ray->dir, vscale(mn, 2.0f * vdot(mn, ray->dir)));
        /* project_to_hemisphere, shader.wgsl:795-801 */
        if (vdot(d, sn) < 0.0f) d = vnormalize(vsub(d, vscale(sn, vdot(d, sn))));
        ray->dir = d;
        ray->idir = vrcp(ray->dir);
    } else { /* sample_lambert, shader.wgsl:781-785 */
        ray->dir = vnormalize(vadd(sn, rand_sphere(cx)));
        ray->idir = vrcp(ray->dir);
    }
}

/* constants of the disabled sun block, shader.wgsl:20-24, 916-937 */
static const float TO_SUN_VAL[3] = {0.5f, 0.4f, 1.49f};

static void render_pixel(Ctx* cx, const OrcFrameUniforms* u, uint32_t px, uint32_t py,
                         uint32_t frame, float out_rgb[3], int32_t* tap_tri, int32_t* tap_prim,
                         float* tap_t) {
    const OrcParams* p = cx->p;
    v3 lighting = V(0.0f, 0.0f, 0.0f), throughput = V(1.0f, 1.0f, 1.0f);
    cx->seed = orc_pixel_seed(px, py, frame);
    cx->debug = 0.0f;
    Ray ray = camera_ray(cx, u, px, py);
    for (uint32_t i = 0; i < p->max_bounces; i++) {
        Hit hit = trace(cx, &ray);
        if (i == 0 && tap_tri) {
            *tap_tri = hit.idx;
            *tap_prim = hit.idx >= 0 ? hit.prim_idx : -1;
            *tap_t = hit.t;
        }
        if (hit.idx == -1) {
            cx->c.misses++;
            lighting = vadd(lighting, vmul(throughput, sky(cx, ray.dir)));
            break;
        }
        if (rnd(cx) < p->term_prob) {
            throughput = vdivs(throughput, p->term_prob);
            cx->c.kills++;
            if (p->enable_sun) {
                /* opt-in extension following the commented-out block: one sun shadow ray */
                float sl = sqrtf(TO_SUN_VAL[0] * TO_SUN_VAL[0] + TO_SUN_VAL[1] * TO_SUN_VAL[1] +
                                 TO_SUN_VAL[2] * TO_SUN_VAL[2]);
                v3 sun = V(TO_SUN_VAL[0] / sl, TO_SUN_VAL[1] / sl, TO_SUN_VAL[2] / sl);
                Ray sr;
                sr.origin = vadd(ray.origin, vadd(vscale(ray.dir, hit.t), vscale(hit.normal, 0.0001f)));
                sr.dir = vnormalize(vadd(sun, vscale(rand_sphere(cx), 0.01f)));
                sr.idir = vrcp(sr.dir);
                float lambert = fmaxf(vdot(sun, hit.normal), 0.0f) / PI_F;
                if (lambert > 0.0f && !trace_shadow(cx, &sr)) {
                    v3 sun_col = V(1.0f * 5.0f, 0.5f * 5.0f, 0.3f * 5.0f);
                    lighting = vadd(lighting, vscale(vscale(vmul(throughput, sun_col), 12.0f), lambert));
                }
            }
            break;
        }
        throughput = vdivs(throughput, 1.0f - p->term_prob);
        surface_hit_brdf(cx, &ray, &hit, &throughput, &lighting);
    }
    if (cx->debug < 0.0f) lighting = V(1.0f, 1.0f, 0.0f);
    out_rgb[0] = lighting.x;
    out_rgb[1] = lighting.y;
    out_rgb[2] = lighting.z;
}

static void counters_add(OrcCounters* a, const OrcCounters* b) {
    uint64_t* pa = (uint64_t*)a;
    const uint64_t* pb = (const uint64_t*)b;
    for (size_t i = 0; i < sizeof(OrcCounters) / sizeof(uint64_t); i++) pa[i] += pb[i];
}

void orc_render(const OrcScene* s, const OrcFrameUniforms* u, const OrcParams* p,
                uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride, float* accum,
                int32_t* tap_tri, int32_t* tap_prim, float* tap_t, uint32_t* tap_seed,
                OrcCounters* counters, int n_threads) {
    orc_render_rows(s, u, p, first_frame, n_frames, frame_stride, accum, tap_tri, tap_prim, tap_t,
                    tap_seed, counters, n_threads, 0, 1);
}

void orc_render_rows(const OrcScene* s, const OrcFrameUniforms* u, const OrcParams* p,
                     uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride, float* accum,
                     int32_t* tap_tri, int32_t* tap_prim, float* tap_t, uint32_t* tap_seed,
                     OrcCounters* counters, int n_threads, uint32_t row_start, uint32_t row_step) {
    const uint32_t W = u->res[0], H = u->res[1];
    if (row_step == 0) row_step = 1;
    OrcCounters total;
    memset(&total, 0, sizeof total);
    if (n_threads < 1) n_threads = 1;
    for (uint32_t f = 0; f < n_frames; f++) {
        const uint32_t frame = first_frame + f * frame_stride;
        const int overwrite = (f == 0 && u->reject_hist == 1);
        const int tap = (f == 0);
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
        {
            Ctx cx;
            memset(&cx, 0, sizeof cx);
            cx.s = s;
            cx.p = p;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
            for (int64_t y = (int64_t)row_start; y < (int64_t)H; y += (int64_t)row_step) {
                for (uint32_t x = 0; x < W; x++) {
                    uint64_t pix = (uint64_t)x + (uint64_t)W * (uint64_t)y;
                    float rgb[3];
                    int32_t tt = -1, tp = -1;
                    float th = 0.0f;
                    render_pixel(&cx, u, x, (uint32_t)y, frame, rgb,
                                 (tap && tap_tri) ? &tt : NULL, &tp, &th);
                    if (tap && tap_tri) { tap_tri[pix] = tt; tap_prim[pix] = tp; tap_t[pix] = th; }
                    if (tap && tap_seed) tap_seed[pix] = cx.seed;
                    float* a = &accum[pix * 4];
                    if (overwrite) { a[0] = rgb[0]; a[1] = rgb[1]; a[2] = rgb[2]; a[3] = 1.0f; }
                    else { a[0] += rgb[0]; a[1] += rgb[1]; a[2] += rgb[2]; a[3] += 1.0f; }
                }
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            counters_add(&total, &cx.c);
        }
    }
    if (counters) *counters = total;
}

int32_t orc_trace_ray(const OrcScene* s, const OrcParams* p, const float origin[3],
                      const float dir[3], uint32_t* seed, float* t_out, int32_t* prim_out) {
    Ctx cx;
    memset(&cx, 0, sizeof cx);
    cx.s = s;
    cx.p = p;
    cx.seed = seed ? *seed : 0u;
    Ray r;
    r.origin = V(origin[0], origin[1], origin[2]);
    r.dir = V(dir[0], dir[1], dir[2]);
    r.idir = vrcp(r.dir);
    Hit h = trace(&cx, &r);
    if (seed) *seed = cx.seed;
    if (t_out) *t_out = h.t;
    if (prim_out) *prim_out = h.idx >= 0 ? h.prim_idx : -1;
    return h.idx;
}

/* ------------------------------------------------------------------ present pass */

/* shader.wgsl:969-992 */
void orc_tonemap_pbr_neutral(const float in[3], float out[3]) {
    float c[3] = {in[0], in[1], in[2]};
    const float start_compression = 0.76f; /* 0.8 - 0.04 */
    const float desaturation = 0.15f;
    float x = fminf(c[0], fminf(c[1], c[2]));
    float offset = 0.04f;
    if (x < 0.08f) offset = x - 6.25f * x * x;
    for (int k = 0; k < 3; k++) c[k] -= offset;
    float peak = fmaxf(c[0], fmaxf(c[1], c[2]));
    if (peak < start_compression) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return; }
    float d = 1.0f - start_compression;
    float new_peak = 1.0f - d * d / (peak + d - start_compression);
    float sc = new_peak / peak;
    for (int k = 0; k < 3; k++) c[k] *= sc;
    float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    for (int k = 0; k < 3; k++) out[k] = mixf(c[k], new_peak * 1.0f, g);
}

/* shader.wgsl:1001-1011; the render target is sRGB (main.rs:116), so the hardware applies the
 * linear->sRGB transfer and rounds to 8-bit UNORM */
void orc_resolve(const float* accum, uint64_t n_pixels, uint8_t* rgba8, int apply_srgb_oetf) {
    for (uint64_t i = 0; i < n_pixels; i++) {
        const float* a = &accum[i * 4];
        float col[3] = {a[0] / a[3] * 1.0f, a[1] / a[3] * 1.0f, a[2] / a[3] * 1.0f};
        float tm[3];
        orc_tonemap_pbr_neutral(col, tm);
        for (int k = 0; k < 3; k++) {
            float c = tm[k];
            if (!(c > 0.0f)) c = 0.0f;
            if (c > 1.0f) c = 1.0f;
            if (apply_srgb_oetf)
                c = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
            rgba8[i * 4 + k] = (uint8_t)(c * 255.0f + 0.5f);
        }
        rgba8[i * 4 + 3] = 255;
    }
}
