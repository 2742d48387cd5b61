// Internal declarations shared by the host-side scene pipeline sources.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rthost.h"

namespace rthost {

// Byte-exact records of the host<->device contract (scene.rs:713-717, 103-117, 840-850,
// 73-78, 142-165, 181-189; input.rs:20-32).
struct Tri { float v[3][4]; };
struct VertexExt { float tc[2][2]; float normal[2]; uint32_t color; float pad; };
struct TriExt { VertexExt v[3]; };
struct Node { float aabb[6]; uint32_t first; uint32_t count; };
struct TexRef { uint32_t offset; uint32_t size; };
struct Material {
    TexRef   albedo, emissive, normal, metallic_roughness;
    float    albedo_factor[4];
    float    emissive_factor[3];
    float    normal_scale;
    uint32_t albedo_texcoord, emissive_texcoord, normal_texcoord, metal_r_texcoord;
    float    metallic_factor, roughness_factor;
    uint32_t id, alpha_settings;
};
struct Primitive {
    float    transform[16];
    float    inv_transform[16];
    Material material;
    uint32_t bvh_idx;
    uint32_t pad[3];
};
struct GpuCamera {
    float dir[3], fovy, origin[3], focus, aperture, exposure, bloom, dispersion;
};
static_assert(sizeof(Tri) == 48 && sizeof(TriExt) == 96 && sizeof(Node) == 32, "layout");
static_assert(sizeof(Material) == 96 && sizeof(Primitive) == 240 && sizeof(GpuCamera) == 48, "layout");

struct Scene {
    std::vector<Tri>       tris;
    std::vector<TriExt>    tri_exts;
    std::vector<Node>      nodes;
    std::vector<Primitive> prims;
    std::vector<uint32_t>  texels;
    std::vector<float>     env;
    uint32_t               env_w = 0, env_h = 0;
    std::vector<uint32_t>  tri_order;
    std::vector<rthost_camera> cameras;
    int                    build_threads = 0;
};

void set_error(const std::string& msg);

uint32_t bvh_build(const float* tris, uint32_t offset, uint32_t size, Node* nodes_out,
                   uint32_t* indices_out, int n_threads);

// scene.cpp
int add_primitive(Scene& s, const float* positions, uint32_t n_verts, const uint32_t* indices,
                  uint64_t n_indices, const float* normals, const float* uv0, const float* uv1,
                  const uint32_t* colors, const Material& material, const float transform[16],
                  uint32_t* out_prim, bool normals_are_unit = false);
int add_instance(Scene& s, uint32_t prim, const float transform[16], uint32_t* out_prim);
int add_texture(Scene& s, const uint8_t* rgba8, uint32_t w, uint32_t h, TexRef* out);
void mat4_mul(const float a[16], const float b[16], float out[16]);
void from_gltf_mat4(const float gltf[16], float out[16]);
void mat4_inverse_f64(const float m[16], float out[16]);
void mat4_inverse_f32(const float m[16], float out[16]);
void pack_oct(const float n[3], float out[2]);
uint32_t pack_alpha(uint32_t mode, float cutoff);

// gltf.cpp / png.cpp / jpeg.cpp / hdr.cpp
int load_gltf(Scene& s, const uint8_t* bytes, size_t n, const std::string& base_dir);
bool decode_png(const uint8_t* bytes, size_t n, std::vector<uint8_t>& rgba8, uint32_t& w,
                uint32_t& h, std::string& err);
bool decode_jpeg(const uint8_t* bytes, size_t n, std::vector<uint8_t>& rgba8, uint32_t& w,
                 uint32_t& h, std::string& err);
bool decode_hdr(const uint8_t* bytes, size_t n, std::vector<float>& rgba32f, uint32_t& w,
                uint32_t& h, std::string& err);
bool read_file(const std::string& path, std::vector<uint8_t>& out);

} // namespace rthost

// the opaque handle of the C ABI
struct rthost_scene { rthost::Scene s; };
