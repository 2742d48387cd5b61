/*
 * rthip.h -- C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference has no plugin or FFI interface; its de-facto boundary is the wgpu binding
 * interface of src/shader.wgsl (lines 1-9) driven by two host functions in src/main.rs:
 * upload_scene (328-353) and frame (356-407).  Each entry point below names the piece of that
 * boundary it replaces.  Everything is plain C: pointers, sizes, int status.  No exceptions or
 * panics cross the ABI.
 *
 * Status: 0 = OK, negative = error; rthip_last_error(ctx) returns the text (ctx may be NULL
 * for errors of rthip_create / rthip_device_count).
 * Threading: a context is bound to one HIP device and is NOT thread-safe; different contexts
 * may be driven from different threads or processes.
 * Ownership: input pointers are borrowed for the duration of the call and copied to the
 * device; outputs are caller-allocated host buffers unless a parameter says "device".
 * Streams: every context owns one HIP stream; all work is enqueued there and calls that return
 * data synchronise it.  rthip_set_stream substitutes a caller's stream.  A context that renders
 * small frame groups also makes one non-blocking stream and three events of its own, at the first
 * such call: a small kernel that only prepares later calls runs there, behind the call's launch and
 * ordered against the calls by events, so the caller's synchronisation of its stream does not
 * wait for it; rthip_set_stream and rthip_destroy do.
 */
#ifndef RTHIP_H
#define RTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RTHIP_OK = 0,
    RTHIP_ERR_INVALID = -1,    /* bad argument (null pointer, size mismatch, index out of range) */
    RTHIP_ERR_HIP = -2,        /* a HIP runtime call failed; text carries hipGetErrorString */
    RTHIP_ERR_STATE = -3,      /* call order: render before upload_scene / set_frame_uniforms */
    RTHIP_ERR_NO_DEVICE = -4,  /* no such device ordinal */
    RTHIP_ERR_UNSUPPORTED = -5 /* a parameter value this build does not implement */
} rthip_status;

typedef struct rthip_ctx rthip_ctx;

/* The six storage buffers and the environment texture the reference binds at group 1
 * (shader.wgsl:3-9), as host arrays in the reference's byte layouts; what
 * Context::upload_scene writes with queue.write_buffer / write_texture (main.rs:330-349). */
typedef struct {
    const void*     tris;      uint64_t n_tris;     /* 48 B each  (scene.rs:713-717)  */
    const void*     tri_exts;                       /* 96 B each  (scene.rs:113-117)  */
    const void*     nodes;     uint64_t n_nodes;    /* 32 B each  (scene.rs:840-850)  */
    const void*     prims;     uint32_t n_prims;    /* 240 B each (scene.rs:181-189)  */
    const uint32_t* texels;    uint64_t n_texels;   /* scene.rs:239, 290-297          */
    const float*    env;       uint32_t env_w, env_h; /* rgba32f, scene.rs:251, 603-615 */
} rthip_scene_desc;

/* The reference compiles these into the shader (shader.wgsl:898, 896, 765, 303); the defaults
 * set by rthip_default_params reproduce it exactly. */
typedef struct {
    uint32_t max_bounces;      /* 8: iterations of the bounce loop                         */
    float    term_prob;        /* 0.5: unconditional termination probability after a hit   */
    float    aperture_radius;  /* 0.025 world units (camera.aperture is ignored upstream)  */
    uint32_t stack_depth;      /* 23: traversal stack entries; a push beyond it is dropped
                                  and the pixel painted (1,1,0), shader.wgsl:306-313, 941  */
    uint32_t enable_sun;       /* 0: the live shader has no sun.  1: the block it keeps
                                  commented out (shader.wgsl:916-937): a terminated path sends
                                  one shadow ray (trace_shadow, :528-575) to the sun        */
    uint32_t flags;            /* RTHIP_FLAG_* */
} rthip_render_params;

#define RTHIP_FLAG_COUNTERS 1u   /* maintain rthip_counters (small cost)                       */
#define RTHIP_FLAG_TAPS     2u   /* record primary hits + final seeds of the first frame       */
#define RTHIP_FLAG_TIMING   4u   /* bracket every stage launch with HIP events (rthip_timing)  */
#define RTHIP_FLAG_STAGED   8u   /* small frame groups (up to 8 * 2^20 camera paths) too go through the staged pipeline
                                    (generate / cull / extend / shade launches per bounce) instead of
                                    the single launch in which a lane owns a whole path; the result
                                    is the same bit for bit                                       */

#define RTHIP_MAX_STACK_DEPTH 64u

typedef struct {
    uint64_t rays;            /* trace() calls (shader.wgsl:598): the unit of the Mrays/s metric */
    uint64_t prims_visited;   /* primitives whose root box was tested in local space            */
    uint64_t inner_steps;     /* inner-node steps (two child boxes each), shader.wgsl:496-521   */
    uint64_t leaf_steps;      /* leaf-node steps, shader.wgsl:458-495                           */
    uint64_t tri_tests;       /* ray/triangle tests, shader.wgsl:462                            */
    uint64_t alpha_tests;     /* in-traversal alpha lookups, shader.wgsl:464-487                */
    uint64_t texel_fetches;   /* texture_data reads, traversal + shading                        */
    uint64_t shade_calls;     /* handle_surface_hit_brdf calls, shader.wgsl:912                 */
    uint64_t misses;          /* handle_miss calls, shader.wgsl:904                             */
    uint64_t kills;           /* paths ended by the termination draw, shader.wgsl:907-909       */
    uint64_t stack_overflows; /* dropped pushes                                                 */
    uint64_t paths;           /* camera rays generated                                          */
    /* wave-level efficiency of the extend stage: how many times a wave executed each kind of
     * step (64 lanes each); lane steps / (rounds * 64) is the SIMD utilisation of that step   */
    uint64_t rounds_prim;     /* primitive-entry steps; lane steps = prims_visited              */
    uint64_t rounds_inner;    /* inner-node steps;      lane steps = inner_steps                */
    uint64_t rounds_leaf;     /* triangle-test steps;   lane steps = tri_tests                  */
    uint64_t batches;         /* 64-ray batches pulled from the queue                           */
    uint64_t shadow_rays;     /* enable_sun only: trace_shadow calls, shader.wgsl:630-638       */
} rthip_counters;

#define RTHIP_STAGE_GENERATE 0   /* camera rays; no launch of its own where the cull stage makes them */
#define RTHIP_STAGE_EXTEND   1
#define RTHIP_STAGE_SHADE    2
#define RTHIP_STAGE_RESOLVE  3
#define RTHIP_STAGE_ACCUMULATE 4 /* frame-ordered sum of the per-frame results into screen  */
#define RTHIP_STAGE_SHADOW   5   /* enable_sun only: the shadow rays of a bounce             */
#define RTHIP_STAGE_CULL     6   /* candidate primitives per ray (scenes of up to 128 primitives) */
#define RTHIP_STAGE_PATH     7   /* small frame groups: the whole of cs_main in one launch */
#define RTHIP_NUM_STAGES     8

typedef struct {
    double   ms[RTHIP_NUM_STAGES];        /* summed kernel time per stage, HIP events       */
    uint64_t launches[RTHIP_NUM_STAGES];  /* launches per stage                             */
} rthip_timing;

int  rthip_device_count(void);
/* Gpu::new + Context::init's buffer creation (gpu.rs:508-590, main.rs:200-221). */
int  rthip_create(int device_ordinal, rthip_ctx** out);
void rthip_destroy(rthip_ctx* ctx);
const char* rthip_last_error(const rthip_ctx* ctx);

/* All later work is enqueued on `hip_stream` (a hipStream_t); NULL restores the context's own
 * stream.  The caller keeps the stream alive.  No counterpart upstream (single wgpu queue). */
int rthip_set_stream(rthip_ctx* ctx, void* hip_stream);

/* Context::upload_scene (main.rs:328-353).  Copies and repacks: nodes keep their 32-byte
 * records and tree, shifted so that sibling pairs are 64-byte aligned; triangles become
 * (p0, e1, e2) records in the uploaded order. */
int rthip_upload_scene(rthip_ctx* ctx, const rthip_scene_desc* scene);

/* queue.write_buffer(frame_uniforms_buffer) (main.rs:380): the exact 608 bytes of
 * FrameUniforms.  Read: scene.camera, res, reject_hist, prim_count; `frame` is overridden by
 * rthip_render's frame arguments.  A change of res re-allocates the accumulator
 * (update_resolution, main.rs:79-94) and implies reject_hist = 1. */
int rthip_set_frame_uniforms(rthip_ctx* ctx, const void* blob608);

void rthip_default_params(rthip_render_params* p);

/* The compute pass of frame() (main.rs:383-393) run n_frames times: frames first_frame,
 * first_frame + frame_stride, ...  The frame index is the only per-sample RNG key
 * (shader.wgsl:895); the reference's first dispatch uses frame 1 (main.rs:372).  If
 * reject_hist is 1 the first frame overwrites the accumulator, and reject_hist is then cleared
 * (main.rs:403).  Asynchronous: returns once the work is enqueued. */
int rthip_render(rthip_ctx* ctx, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                 const rthip_render_params* params);

/* Allocates the per-frame work buffers for calls of up to n_frames frames now rather than inside
 * the first rthip_render that needs them (frames travel through the stages in groups; a larger
 * group needs larger queues).  Optional.  No counterpart upstream: wgpu buffers are created once
 * per resolution (main.rs:79-94). */
int rthip_reserve(rthip_ctx* ctx, uint32_t n_frames);

/* How many frames the next rthip_render would move through the stages together at the current
 * resolution: RTHIP_FRAMES_IN_FLIGHT (default 128), lowered until the work buffers (160 bytes per
 * path slot) fit 96 GiB and half of the device memory that is free -- a limit that is otherwise
 * applied silently.  No counterpart upstream. */
int rthip_frames_in_flight(rthip_ctx* ctx, uint32_t* n_frames);

/* How the extend stage finds the primitives a ray may touch in the uploaded scene (the reference
 * tests every primitive for every ray, shader.wgsl:603): RTHIP_WALK_MASK = up to 128 primitives, one
 * bit mask per ray; RTHIP_WALK_SCAN = chunks of 128 primitives scanned wave by wave;
 * RTHIP_WALK_PYRAMID = per-ray walk of an 8-ary pyramid of bounds over the primitive list, chosen
 * for long lists whose neighbours are neighbours in space; RTHIP_WALK_LIST = per-ray candidate lists,
 * sorted by index, from a spatial tree over the primitives' bounds (rthip_trace_rays_noalloc uses
 * the scan for such a scene), chosen for long lists in an order that says nothing about space.  The
 * result of a render does not depend on it.  No counterpart upstream. */
#define RTHIP_WALK_MASK 0
#define RTHIP_WALK_SCAN 1
#define RTHIP_WALK_PYRAMID 2
#define RTHIP_WALK_LIST 3
int rthip_candidate_walk(rthip_ctx* ctx, uint32_t* kind);

/* Use caller-owned DEVICE memory (n_pixels * 4 floats, 16-byte aligned) as the accumulator,
 * e.g. a tensor that a collective library will reduce in place.  NULL returns to a
 * library-owned buffer.  The `screen` storage buffer of shader.wgsl:6. */
int rthip_bind_accum(rthip_ctx* ctx, void* device_ptr, uint64_t n_pixels);
int rthip_accum_device_ptr(rthip_ctx* ctx, void** device_ptr, uint64_t* n_pixels);

/* Read back / restore `screen` (rgb = radiance sum, a = sample count).  The reference never
 * reads it back; its buffers carry COPY_SRC (gpu.rs:490-493). */
int rthip_read_accum(rthip_ctx* ctx, float* rgba, uint64_t n_pixels);
int rthip_write_accum(rthip_ctx* ctx, const float* rgba, uint64_t n_pixels);

/* Parity taps of the FIRST frame of the last rthip_render issued with RTHIP_FLAG_TAPS: per
 * pixel the primary hit's triangle index (-1 = miss), primitive index, distance, and the RNG
 * state when the path ended.  Any pointer may be NULL. */
int rthip_read_taps(rthip_ctx* ctx, int32_t* tri_idx, int32_t* prim_idx, float* t,
                    uint32_t* final_seed, uint64_t n_pixels);

/* Proof that the device holds the identical tree: the device node array (32-byte records;
 * capacity in records) and, per uploaded node, its device index. */
int rthip_read_nodes_device_layout(rthip_ctx* ctx, void* nodes_out, uint64_t capacity,
                                   uint64_t* n_records);
int rthip_read_node_index_map(rthip_ctx* ctx, uint32_t* device_index_of_node, uint64_t n_nodes);
/* Order-preserving triangle map: device triangle i is uploaded triangle out[i]. */
int rthip_read_tri_order(rthip_ctx* ctx, uint32_t* out, uint64_t n_tris);
/* The device triangle records (48 bytes each: p0.xyz,0, e1.xyz,0, e2.xyz,0). */
int rthip_read_tris_device_layout(rthip_ctx* ctx, void* out, uint64_t n_tris);

/* trace() (shader.wgsl:598-627) for caller-supplied rays: the closest hit of each of n rays
 * through every primitive, exactly as a path segment finds it.  rays6 holds origin.xyz and
 * direction.xyz per ray (world space; the direction is used as given).  seeds (may be NULL: all
 * zero) are the RNG states the BLEND alpha test draws from and come back advanced.  Outputs (any
 * may be NULL): distance (99999999 on a miss), triangle index (-1 on a miss), primitive index,
 * and 1 where a traversal-stack push was dropped.  Scene::focus_camera (scene.rs:670-677) is one
 * such ray.  Synchronous. */
int rthip_trace_rays(rthip_ctx* ctx, const float* rays6, uint32_t* seeds, uint64_t n,
                     const rthip_render_params* params, float* t, int32_t* tri_idx,
                     int32_t* prim_idx, uint8_t* overflow);

/* rthip_trace_rays for callers that cannot allocate.  No counterpart in the reference: it is for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Calls no
 * hipMalloc / hipFree (nor any allocating library routine): the kernels' work memory (queue
 * planes, hit records, control words, deep traversal-stack columns) is a fixed array in the
 * library's device image, about 19 MB that every GPU the library runs on holds until the process
 * ends.  At most RTHIP_NOALLOC_MAX_RAYS rays per call; more is RTHIP_ERR_INVALID.
 * ALL pointers are DEVICE memory of the context's GPU (caller-owned): d_rays6 as rays6 above,
 * d_seeds (may be NULL) read and advanced in place, outputs as above (any may be NULL).
 * Asynchronous: the work is enqueued on the context's stream and nothing is synchronised.
 * Every call on a GPU, from any context, uses that GPU's one array.  The library orders them: a call
 * waits, on its stream, for the previous call on that GPU (one event per GPU, created with the first
 * context on it; no allocation per call), so two contexts or two streams may issue calls freely.
 * The exception is a stream under graph capture, where the event is not used: calls captured into
 * graphs that may run at the same time on one GPU race on the array, and keeping them apart is the
 * caller's business. */
#define RTHIP_NOALLOC_MAX_RAYS (1u << 18)
int rthip_trace_rays_noalloc(rthip_ctx* ctx, const float* d_rays6, uint32_t* d_seeds, uint64_t n,
                             const rthip_render_params* params, float* d_t, int32_t* d_tri_idx,
                             int32_t* d_prim_idx, uint8_t* d_overflow);

/* The render pass of frame() (main.rs:395-401; fs_main, shader.wgsl:994-1012): mean, PBR
 * Neutral tonemap, then the linear->sRGB transfer the reference gets from its sRGB surface
 * (main.rs:116) when apply_srgb_oetf != 0, rounded to 8 bits; alpha = 255.  `rgba8` is host
 * memory (pageable or pinned: the image is copied there) or 4-byte-aligned memory of the context's
 * GPU (a presentation buffer: the kernel writes it in place, no copy and no allocation).
 * Returns when the image is there. */
int rthip_resolve(rthip_ctx* ctx, uint8_t* rgba8, uint64_t n_pixels, int apply_srgb_oetf);

int rthip_get_counters(rthip_ctx* ctx, rthip_counters* out);
int rthip_reset_counters(rthip_ctx* ctx);
int rthip_get_timing(rthip_ctx* ctx, rthip_timing* out);
int rthip_reset_timing(rthip_ctx* ctx);

/* Measurement aid: times `repeats` device-to-device copies of `bytes` bytes on the context's
 * stream and returns the achieved HBM traffic in GB/s (bytes read + bytes written per second), the
 * stream-copy ceiling that bandwidth fractions can be quoted against.  No counterpart upstream. */
int rthip_copy_bandwidth(rthip_ctx* ctx, uint64_t bytes, uint32_t repeats, double* gb_per_s);

/* Measurement aid: the rate at which the GPU's vector-memory pipelines serve the access shape of the
 * traversal when memory latency is out of the picture -- `active_lanes` scattered lanes of every
 * wave (1..64) each read the 16-byte pieces of `records_in_flight` (1 or 2) records of their own per
 * step -- record_bytes = 64: four pieces, as an inner-node step of trace_bvh (shader.wgsl:496-521)
 * reads its sibling pair; record_bytes = 128: eight, what a step of a 4-wide tree with exact boxes
 * would read -- from a table that stays in the L1, eight waves per SIMD, no dependence between
 * steps.  Returns lane-loads (16 bytes each) per second over the whole chip.  The largest of the
 * shapes is the ceiling that the extend stage's own lane-loads per second are quoted against on
 * scenes that live in the caches, where HBM bandwidth does not bind.  No counterpart upstream. */
int rthip_gather_ceiling(rthip_ctx* ctx, uint32_t active_lanes, uint32_t record_bytes, uint32_t records_in_flight, double* lane_loads_per_s);

int rthip_synchronize(rthip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
