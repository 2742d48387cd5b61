/*
 * rthost.h -- C ABI of the host-side scene pipeline that FEEDS the path-tracing hot path.
 *
 * It stands where the reference's Rust host stands (src/scene.rs, src/input.rs): it turns
 * meshes, textures, an environment map and a camera into exactly the flat byte arrays the
 * reference uploads to its shader (main.rs:328-353) and the 608-byte FrameUniforms block it
 * rewrites every frame (main.rs:372-380).  No GPU code lives behind this header; the device
 * side is rthip.h.  A Rust host that keeps its own scene.rs does not need this library at
 * all -- it hands its Vec<Tri>/Vec<TriExt>/Vec<BvhNode>/Vec<Primitive>/Vec<u32>/Vec<[f32;4]>
 * straight to rthip_upload_scene.
 *
 * Conventions: every function returns 0 on success or a negative rthost_status;
 * rthost_last_error() returns the message of the calling thread's last failure.  Pointers
 * handed in are borrowed for the duration of the call.  A scene object is not thread-safe.
 * World space is the reference's: X forward, Z up, -Y right (input.rs:6-8); matrices are
 * column-major float[16].
 */
#ifndef RTHOST_H
#define RTHOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RTHOST_OK = 0,
    RTHOST_ERR_INVALID = -1,   /* bad argument */
    RTHOST_ERR_IO = -2,        /* file missing / unreadable */
    RTHOST_ERR_FORMAT = -3,    /* malformed or unsupported glTF / PNG / HDR content */
    RTHOST_ERR_LIMIT = -4      /* a 32-bit index or 16-bit texture dimension would overflow */
} rthost_status;

typedef struct rthost_scene rthost_scene;

/* Byte sizes of the records of the host<->device contract (scene.rs:713-717, 103-117,
 * 840-850, 181-189, 142-165; input.rs:20-32; main.rs:31-43). */
#define RTHOST_TRI_BYTES        48u
#define RTHOST_TRI_EXT_BYTES    96u
#define RTHOST_NODE_BYTES       32u
#define RTHOST_PRIMITIVE_BYTES 240u
#define RTHOST_MATERIAL_BYTES   96u
#define RTHOST_CAMERA_BYTES     48u
#define RTHOST_UNIFORMS_BYTES  608u

/* Borrowed view of a scene's flat arrays; valid until the scene is next modified or freed.
 * The same six arrays the reference binds at group 1 (shader.wgsl:3-9). */
typedef struct {
    const void*     tris;       uint64_t n_tris;     /* 48 B each, BVH leaf order            */
    const void*     tri_exts;                        /* 96 B each, same order                */
    const void*     nodes;      uint64_t n_nodes;    /* 32 B each, all primitives appended   */
    const void*     prims;      uint32_t n_prims;    /* 240 B each                           */
    const uint32_t* texels;     uint64_t n_texels;   /* R<<24|G<<16|B<<8|A, textures appended */
    const float*    env;        uint32_t env_w, env_h; /* rgba32f rows, top row first        */
    const uint32_t* tri_order;  /* tri_order[i] = index, in load order, of the triangle now at i */
    uint32_t        n_cameras;
} rthost_scene_view;

const char* rthost_last_error(void);

/* Scene::default (scene.rs:223-252). */
int  rthost_scene_new(rthost_scene** out);
void rthost_scene_free(rthost_scene* s);
int  rthost_scene_view_get(const rthost_scene* s, rthost_scene_view* out);

/* Scene::add_gltf_texture's packing (scene.rs:290-297, 322-336): rgba8 rows, top row first,
 * appended to the texel array.  out_ref = {offset, w<<16|h} (scene.rs:80-87). */
int rthost_scene_add_texture(rthost_scene* s, const uint8_t* rgba8, uint32_t w, uint32_t h,
                             uint32_t out_ref[2]);

/* What add_gltf_node does for one triangle primitive once its attributes are read
 * (scene.rs:516-577): one Tri + TriExt per index triple, BVH build over the new triangles,
 * flatten, append + rebase nodes, Primitive::new with the f64 inverse.
 * positions: n_verts x 3 floats in world axes.  normals (optional): n_verts x 3, any length,
 * normalised then octahedral-packed (scene.rs:506-511).  uv0/uv1 (optional): n_verts x 2.
 * colors (optional): n_verts packed rgba8 (R<<24..A).  material: 96 bytes.
 * out_prim receives the primitive index. */
int rthost_scene_add_primitive(rthost_scene* s, const float* positions, uint32_t n_verts,
                               const uint32_t* indices, uint64_t n_indices, const float* normals,
                               const float* uv0, const float* uv1, const uint32_t* colors,
                               const void* material96, const float transform[16],
                               uint32_t* out_prim);

/* Instancing (scene.rs:379-384): copy of primitive `prim` with a new transform and the f32
 * inverse; shares triangles, nodes and material. */
int rthost_scene_add_instance(rthost_scene* s, uint32_t prim, const float transform[16],
                              uint32_t* out_prim);

/* Scene::set_equirectangular_env_map's result (scene.rs:603-615): w*h rgba32f texels. */
int rthost_scene_set_env(rthost_scene* s, const float* rgba32f, uint32_t w, uint32_t h);

/* Scene::add_gltf_bytes (scene.rs:268-284) over a .gltf (JSON, data: URIs or files relative to
 * base_dir) or .glb container; Scene::from_path (scene.rs:680-708) = new + load_gltf +
 * load_hdr + default camera when the file has none. */
int rthost_scene_load_gltf(rthost_scene* s, const void* bytes, size_t n_bytes,
                           const char* base_dir);
int rthost_scene_load_gltf_file(rthost_scene* s, const char* path);
int rthost_scene_load_hdr_file(rthost_scene* s, const char* path);

/* .rtscene cache: every flat array of the scene (and its cameras) in one file, written once and
 * then only read.  For scenes whose source needs a decoder this build lacks (CMYK JPEG, ...) or
 * that take long to build; the reference's Rust host can write the same file because its Vecs
 * hold the same bytes (scene.rs:239-251).  load_cache replaces the scene's content and refuses
 * files whose lengths or tree indices are inconsistent.  Layout: csrc/host/cache.cpp. */
int rthost_scene_save(const rthost_scene* s, const char* path);
int rthost_scene_load_cache(rthost_scene* s, const char* path);

/* Cameras (input.rs:34-51, 202-253).  A camera is yaw/pitch/roll + position + fovy + focus. */
typedef struct {
    float yaw, pitch, roll;
    float position[3];
    float fovy, focus;
    float aperture, exposure, bloom, dispersion;
    float aspect;      /* 0 when the source has none */
} rthost_camera;

void rthost_camera_default(rthost_camera* out);                       /* input.rs:202-219 */
void rthost_camera_to_gpu(const rthost_camera* cam, void* out48);     /* input.rs:100-111 */
int  rthost_scene_add_camera(rthost_scene* s, const rthost_camera* cam);
int  rthost_scene_get_camera(const rthost_scene* s, uint32_t idx, rthost_camera* out);
/* Scene::focus_camera (scene.rs:640-677): CPU ray cast along the camera forward axis. */
int  rthost_scene_focus_camera(rthost_scene* s, uint32_t idx, int* hit);

/* FrameUniforms as frame() writes it (main.rs:183-192, 372-380; scene.rs:617-638): lights
 * zeroed, camera idx (explicit 48-byte camera if gpu_camera48 != NULL), res, frame, time,
 * reject_hist, node_count, prim_count. */
int rthost_frame_uniforms(const rthost_scene* s, uint32_t camera_idx, const void* gpu_camera48,
                          uint32_t res_x, uint32_t res_y, uint32_t frame, float time,
                          uint32_t reject_hist, void* out608);

/* The BVH builder alone (Bvh::new, scene.rs:888-899, 991-1050): builds over
 * tris[offset, offset+size) of a 48-byte triangle array.  nodes_out needs room for 2*size
 * records; indices_out for `size` global triangle ids (the permutation flatten_triangles
 * applies).  Node indices are local to this BVH (root = 0).  n_threads <= 0 picks a default.
 * Returns the node count through n_nodes_out.  The result is bit-identical to the reference
 * builder's for every n_threads. */
int rthost_bvh_build(const void* tris48, uint32_t offset, uint32_t size, void* nodes_out,
                     uint32_t* indices_out, uint32_t* n_nodes_out, int n_threads);

/* Helpers exposed for tests and foreign hosts (scene.rs:170-177, 203-221, 720-725;
 * input.rs:67-76). */
void     rthost_tri_new(const float p1[3], const float p2[3], const float p3[3], void* out48);
void     rthost_pack_oct(const float n[3], float out[2]);
uint32_t rthost_pack_alpha(uint32_t mode, float cutoff);
void     rthost_from_gltf_mat4(const float gltf[16], float out[16]);
void     rthost_mat4_inverse_f64(const float m[16], float out[16]);   /* scene.rs:195 */
void     rthost_mat4_inverse_f32(const float m[16], float out[16]);   /* scene.rs:382 */

#ifdef __cplusplus
}
#endif
#endif
