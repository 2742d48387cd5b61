/*
 * rt_oracle.h -- CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the algorithm in the reference's src/shader.wgsl (compute entry
 * cs_main and the present pass fs_main) and of the BVH builder in src/scene.rs.  It exists so
 * that tests, the smoke check and the benchmark's CPU-baseline leg can check / time the HIP
 * path against an independent scalar implementation.  Nothing that ships may include, link or
 * call this code.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, and neither its Rust
 * host nor its WGSL shader can be built or run in the authoring environment.  The oracle is
 * pinned only by (i) hand-derived integer known-answer vectors for hash21/rand (exact u32
 * arithmetic from shader.wgsl:641-659) and (ii) inspection against the cited source lines.
 *
 * All struct layouts below are the byte-exact host<->device contract of the reference
 * (scene.rs:9-22,61-78,103-117,142-165,181-201,713-717,766-771,840-850; input.rs:20-32;
 * main.rs:31-43; mirrored in shader.wgsl:28-125,162-167,200-202,260-292).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scene.rs:713-717 / shader.wgsl:285-292: (p0.xyz,c.x)(p1.xyz,c.y)(p2.xyz,c.z) */
typedef struct { float v[3][4]; } OrcTri;                                   /* 48 B */
/* scene.rs:103-110 / shader.wgsl:162-167 */
typedef struct { float tc[2][2]; float normal[2]; uint32_t color; float pad; } OrcVertexExt; /* 32 B */
typedef struct { OrcVertexExt v[3]; } OrcTriExt;                            /* 96 B */
/* scene.rs:840-850 / shader.wgsl:274-282 */
typedef struct { float aabb[6]; uint32_t first; uint32_t count; } OrcNode;  /* 32 B */
/* scene.rs:73-78 */
typedef struct { uint32_t offset; uint32_t size; } OrcTexRef;               /* 8 B */
/* scene.rs:142-165 / shader.wgsl:54-75 */
typedef struct {
    OrcTexRef albedo, emissive, normal, metallic_roughness;
    float     albedo_factor[4];
    float     emissive_factor[3];
    float     normal_scale;
    uint32_t  albedo_texcoord, emissive_texcoord, normal_texcoord, metal_r_texcoord;
    float     metallic_factor, roughness_factor;
    uint32_t  id, alpha_settings;
} OrcMaterial;                                                              /* 96 B */
/* scene.rs:181-189 / shader.wgsl:100-106; matrices column-major */
typedef struct {
    float       transform[16];
    float       inv_transform[16];
    OrcMaterial material;
    uint32_t    bvh_idx;
    uint32_t    pad[3];
} OrcPrimitive;                                                             /* 240 B */
/* input.rs:20-32 / shader.wgsl:28-37 */
typedef struct {
    float dir[3];    float fovy;
    float origin[3]; float focus;
    float aperture, exposure, bloom, dispersion;
} OrcCamera;                                                                /* 48 B */
/* main.rs:31-43 over scene.rs:61-71 / shader.wgsl:108-125 */
typedef struct {
    uint8_t   lights[512];          /* 12 point + 4 directional lights, never read by the path */
    OrcCamera camera;               /* @512 */
    uint32_t  tri_count, num_point_lights, num_directional_lights, pad0;   /* @560 */
    uint32_t  res[2];               /* @576 */
    uint32_t  frame;                /* @584 */
    float     time;                 /* @588 */
    uint32_t  reject_hist;          /* @592 */
    uint32_t  node_count;           /* @596 */
    uint32_t  prim_count;           /* @600 */
    uint32_t  pad1;                 /* @604 */
} OrcFrameUniforms;                                                         /* 608 B */

typedef struct {
    const OrcTri*       tris;      uint64_t n_tris;
    const OrcTriExt*    exts;
    const OrcNode*      nodes;     uint64_t n_nodes;
    const OrcPrimitive* prims;     uint32_t n_prims;
    const uint32_t*     texels;    uint64_t n_texels;
    const float*        env;       uint32_t env_w, env_h;   /* rgba32f row-major */
} OrcScene;

/* The reference hard-codes these (shader.wgsl:898, 896, 765, 303); defaults reproduce it. */
typedef struct {
    uint32_t max_bounces;      /* 8   */
    float    term_prob;        /* 0.5 */
    float    aperture_radius;  /* 0.025 */
    uint32_t stack_depth;      /* 23  */
    uint32_t enable_sun;       /* 0: the live shader has no sun (shader.wgsl:916-937 is commented out) */
} OrcParams;

typedef struct {
    uint64_t rays;            /* trace() calls, shader.wgsl:598 */
    uint64_t root_tests;      /* root slab tests (one per primitive per ray), :448 */
    uint64_t inner_steps;     /* inner-node iterations of the traversal loop, :496-521 */
    uint64_t leaf_steps;      /* leaf-node iterations, :458-495 */
    uint64_t tri_tests;       /* intersect() calls, :462 */
    uint64_t alpha_tests;     /* in-traversal alpha lookups, :464-487 */
    uint64_t texel_fetches;   /* texture_data reads (traversal + shading) */
    uint64_t shade_calls;     /* handle_surface_hit_brdf calls, :912 */
    uint64_t misses;          /* handle_miss calls, :904 */
    uint64_t kills;           /* paths ended by the termination draw, :907-909 */
    uint64_t rng_draws;       /* rand() calls */
    uint64_t stack_overflows; /* dropped pushes, :310-311 */
    uint64_t shadow_rays;     /* sun extension only */
} OrcCounters;

/* ---- integer RNG (shader.wgsl:641-659) ---- */
uint32_t orc_hash21(uint32_t x, uint32_t y);
/* advances *seed, returns the raw u32 that rand() converts to float */
uint32_t orc_rand_u32(uint32_t* seed);
float    orc_rand(uint32_t* seed);
uint32_t orc_pixel_seed(uint32_t px, uint32_t py, uint32_t frame);

/* ---- BVH builder (scene.rs:853-1050) ----
 * tris: the scene-wide triangle array; builds over [offset, offset+size).
 * nodes_out: capacity >= 2*size.  indices_out: size entries (global triangle ids, the
 * permutation captured before flatten_triangles resets it).  Returns the node count. */
uint32_t orc_bvh_build(const OrcTri* tris, uint32_t offset, uint32_t size,
                       OrcNode* nodes_out, uint32_t* indices_out);
/* scene.rs:553-564: build + flatten (reorders tris/exts in place) + rebase inner nodes by
 * bvh_root.  nodes_out points at the slot for scene-wide node index bvh_root.
 * order_out (optional): order_out[i] = original index (relative to offset) now at offset+i. */
uint32_t orc_build_primitive(OrcTri* tris, OrcTriExt* exts, uint32_t offset, uint32_t size,
                             OrcNode* nodes_out, uint32_t bvh_root, uint32_t* order_out);
/* scene.rs:720-725 */
void orc_tri_new(const float p1[3], const float p2[3], const float p3[3], OrcTri* out);

/* ---- host packing helpers (scene.rs:170-177, 203-221, 192-200; input.rs:67-111, 202-219) ---- */
void     orc_pack_oct(const float n[3], float out[2]);
uint32_t orc_pack_alpha(uint32_t mode, float cutoff);
void     orc_mat4_inverse_f64(const float m[16], float out[16]);
void     orc_camera_default(OrcCamera* out);
void     orc_camera_dir(float yaw, float pitch, float out[3]);

/* ---- the path (shader.wgsl:886-953) ----
 * Renders frames first_frame, first_frame+stride, ... (n_frames of them) into accum
 * (res.x*res.y*4 floats; rgb = sum of radiance, a = sample count).  u->reject_hist == 1 makes
 * the first rendered frame overwrite (shader.wgsl:947-951), later frames add (main.rs:403).
 * Optional taps (may be NULL), all for the FIRST rendered frame: primary-hit triangle index,
 * primitive index and t per pixel; final RNG seed per pixel.
 * n_threads <= 1 runs single-threaded. */
void orc_render(const OrcScene* s, const OrcFrameUniforms* u, const OrcParams* p,
                uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                float* accum, int32_t* tap_tri, int32_t* tap_prim, float* tap_t,
                uint32_t* tap_seed, OrcCounters* counters, int n_threads);

/* Same, restricted to image rows row_start, row_start + row_step, ... (a bounded sample of a
 * full-resolution frame for CPU timing); other rows of accum are left untouched. */
void orc_render_rows(const OrcScene* s, const OrcFrameUniforms* u, const OrcParams* p,
                     uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                     float* accum, int32_t* tap_tri, int32_t* tap_prim, float* tap_t,
                     uint32_t* tap_seed, OrcCounters* counters, int n_threads,
                     uint32_t row_start, uint32_t row_step);

/* one trace() of an explicit world-space ray: returns tri index (-1 = miss) */
int32_t orc_trace_ray(const OrcScene* s, const OrcParams* p, const float origin[3],
                      const float dir[3], uint32_t* seed, float* t_out, int32_t* prim_out);

/* ---- present pass (shader.wgsl:969-1012 + sRGB target, main.rs:116) ---- */
void orc_tonemap_pbr_neutral(const float in[3], float out[3]);
void orc_resolve(const float* accum, uint64_t n_pixels, uint8_t* rgba8, int apply_srgb_oetf);

#ifdef __cplusplus
}
#endif
#endif
