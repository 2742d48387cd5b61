// .rtscene: the flat arrays of a loaded scene in one file, so that a scene whose source needs a
// decoder this build does not have (CMYK JPEG, ...), or that takes long to build, is prepared
// once -- by this library or by the reference's Rust host, whose Vecs have the same bytes
// (scene.rs:239-251) -- and then only read.
//
// Little-endian.  Header (64 bytes):
//   char[8]  "RTSCENE1"
//   u64      n_tris, n_nodes, n_texels
//   u32      n_prims, env_w, env_h, n_cameras
//   u8[16]   zero
// then, each padded to a multiple of 16 bytes:
//   tris (48 B each), tri_exts (96 B), nodes (32 B), prims (240 B), texels (u32),
//   env (rgba32f), tri_order (u32 per triangle), cameras (13 f32 each: rthost_camera).
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "rthost_internal.h"

namespace {

const char MAGIC[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', '1'};

struct Header {
    char     magic[8];
    uint64_t n_tris, n_nodes, n_texels;
    uint32_t n_prims, env_w, env_h, n_cameras;
    uint8_t  zero[16];
};
static_assert(sizeof(Header) == 64, "header");

bool write_block(FILE* f, const void* p, size_t bytes) {
    static const char pad[16] = {0};
    if (bytes && std::fwrite(p, 1, bytes, f) != bytes) return false;
    const size_t rest = (16 - bytes % 16) % 16;
    return rest == 0 || std::fwrite(pad, 1, rest, f) == rest;
}

template <typename T> bool read_block(FILE* f, std::vector<T>& v, uint64_t count) {
    v.resize(count);
    const size_t bytes = sizeof(T) * (size_t)count;
    if (bytes && std::fread(v.data(), 1, bytes, f) != bytes) return false;
    const size_t rest = (16 - bytes % 16) % 16;
    char pad[16];
    return rest == 0 || std::fread(pad, 1, rest, f) == rest;
}

} // namespace

extern "C" int rthost_scene_save(const rthost_scene* sp, const char* path) {
    if (!sp || !path) { rthost::set_error("scene_save: null argument"); return RTHOST_ERR_INVALID; }
    const rthost::Scene& s = sp->s;
    if (s.tri_exts.size() != s.tris.size() || s.tri_order.size() != s.tris.size()) { rthost::set_error("scene_save: inconsistent scene"); return RTHOST_ERR_INVALID; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { rthost::set_error(std::string("scene_save: cannot open ") + path); return RTHOST_ERR_IO; }
    Header h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, MAGIC, 8);
    h.n_tris = s.tris.size();
    h.n_nodes = s.nodes.size();
    h.n_texels = s.texels.size();
    h.n_prims = (uint32_t)s.prims.size();
    h.env_w = s.env_w;
    h.env_h = s.env_h;
    h.n_cameras = (uint32_t)s.cameras.size();
    bool ok = std::fwrite(&h, 1, sizeof h, f) == sizeof h;
    ok = ok && write_block(f, s.tris.data(), sizeof(rthost::Tri) * s.tris.size());
    ok = ok && write_block(f, s.tri_exts.data(), sizeof(rthost::TriExt) * s.tri_exts.size());
    ok = ok && write_block(f, s.nodes.data(), sizeof(rthost::Node) * s.nodes.size());
    ok = ok && write_block(f, s.prims.data(), sizeof(rthost::Primitive) * s.prims.size());
    ok = ok && write_block(f, s.texels.data(), sizeof(uint32_t) * s.texels.size());
    ok = ok && write_block(f, s.env.data(), sizeof(float) * s.env.size());
    ok = ok && write_block(f, s.tri_order.data(), sizeof(uint32_t) * s.tri_order.size());
    ok = ok && write_block(f, s.cameras.data(), sizeof(rthost_camera) * s.cameras.size());
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { rthost::set_error(std::string("scene_save: write to ") + path + " failed"); return RTHOST_ERR_IO; }
    return RTHOST_OK;
}

static int load_cache_body(rthost_scene* sp, const char* path);
extern "C" int rthost_scene_load_cache(rthost_scene* sp, const char* path) {
    try {
        return load_cache_body(sp, path);
    } catch (const std::exception& e) { // an allocation the file's (checked) sizes ask for can still fail
        rthost::set_error(std::string("scene_load_cache: ") + e.what());
        return RTHOST_ERR_LIMIT;
    }
}
static int load_cache_body(rthost_scene* sp, const char* path) {
    if (!sp || !path) { rthost::set_error("scene_load_cache: null argument"); return RTHOST_ERR_INVALID; }
    FILE* f = std::fopen(path, "rb");
    if (!f) { rthost::set_error(std::string("scene_load_cache: cannot open ") + path); return RTHOST_ERR_IO; }
    Header h;
    rthost::Scene t;
    bool ok = std::fread(&h, 1, sizeof h, f) == sizeof h && std::memcmp(h.magic, MAGIC, 8) == 0;
    if (!ok) { std::fclose(f); rthost::set_error(std::string("scene_load_cache: ") + path + " is not an RTSCENE1 file"); return RTHOST_ERR_FORMAT; }
    // sizes are checked against the file length before anything is allocated
    std::fseek(f, 0, SEEK_END);
    const long long file_len = std::ftell(f);
    std::fseek(f, (long)sizeof h, SEEK_SET);
    // header counts are untrusted: bound them first, so that none of the products below can wrap
    // (2^31 records is also what the device library accepts)
    if (h.n_tris >= (1ull << 31) || h.n_nodes >= (1ull << 31) || h.n_texels >= (1ull << 34) || h.env_w > 65536u || h.env_h > 65536u || h.n_cameras > 65536u) {
        std::fclose(f);
        rthost::set_error(std::string("scene_load_cache: ") + path + " has counts beyond what a scene can hold");
        return RTHOST_ERR_FORMAT;
    }
    auto padded = [](unsigned long long b) { return (b + 15ull) & ~15ull; };
    const unsigned long long need = sizeof h + padded(48ull * h.n_tris) + padded(96ull * h.n_tris) + padded(32ull * h.n_nodes) +
                                    padded(240ull * h.n_prims) + padded(4ull * h.n_texels) + padded(16ull * h.env_w * h.env_h) +
                                    padded(4ull * h.n_tris) + padded(sizeof(rthost_camera) * (unsigned long long)h.n_cameras);
    if (file_len < 0 || (unsigned long long)file_len != need) {
        std::fclose(f);
        rthost::set_error(std::string("scene_load_cache: ") + path + " has the wrong length for its header");
        return RTHOST_ERR_FORMAT;
    }
    ok = read_block(f, t.tris, h.n_tris) && read_block(f, t.tri_exts, h.n_tris) && read_block(f, t.nodes, h.n_nodes) &&
         read_block(f, t.prims, h.n_prims) && read_block(f, t.texels, h.n_texels) &&
         read_block(f, t.env, 4ull * h.env_w * h.env_h) && read_block(f, t.tri_order, h.n_tris) &&
         read_block(f, t.cameras, h.n_cameras);
    std::fclose(f);
    if (!ok) { rthost::set_error(std::string("scene_load_cache: short read from ") + path); return RTHOST_ERR_IO; }
    t.env_w = h.env_w;
    t.env_h = h.env_h;
    // the same structural checks an upload would make, so that a damaged file is refused here
    for (const rthost::Primitive& p : t.prims)
        if (p.bvh_idx >= t.nodes.size()) { rthost::set_error("scene_load_cache: primitive root outside the node array"); return RTHOST_ERR_FORMAT; }
    for (size_t i = 0; i < t.nodes.size(); i++) {
        const rthost::Node& n = t.nodes[i];
        const bool bad = n.count > 0 ? (uint64_t)n.first + n.count > t.tris.size() : ((uint64_t)n.first + 1 >= t.nodes.size() || n.first <= i);
        if (bad) { rthost::set_error("scene_load_cache: node " + std::to_string(i) + " points outside its array"); return RTHOST_ERR_FORMAT; }
    }
    rthost::Scene& s = sp->s;
    t.build_threads = s.build_threads;
    s = std::move(t);
    return RTHOST_OK;
}
