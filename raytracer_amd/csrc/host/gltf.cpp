// glTF 2.0 (.gltf JSON or .glb container) -> Scene, with the semantics of Scene::add_gltf_bytes
// and add_gltf_node (scene.rs:268-597) and Camera::from_gltf (input.rs:222-253): scene-graph
// walk with a matrix stack, one primitive per triangle-mode mesh primitive, positions and
// normals re-based to (z, x, y), up to two UV sets (only the sets a texture refers to are
// read), rgba8 vertex colours, material factors + four texture references, alpha mode,
// instancing by (mesh, primitive), perspective cameras.  Punctual lights are skipped: the path
// never reads them (shader.wgsl:39-47 are unused).
#include "rthost_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

namespace rthost {

namespace {

// ---------------------------------------------------------------- minimal JSON
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool   b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char* key) const { return get(key) != nullptr; }
    double number(const char* key, double def) const {
        const Json* j = get(key);
        return j && j->kind == Num ? j->num : def;
    }
    int64_t integer(const char* key, int64_t def) const {
        const Json* j = get(key);
        return j && j->kind == Num ? (int64_t)j->num : def;
    }
    std::string string(const char* key, const std::string& def = "") const {
        const Json* j = get(key);
        return j && j->kind == Str ? j->str : def;
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
};

struct JsonParser {
    const char* p;
    const char* end;
    std::string err;

    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }

    bool parse_string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) out.push_back((char)cp);
                        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        break;
                    }
                    default: out.push_back(*p);
                }
                p++;
            } else {
                out.push_back(*p++);
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }

    bool parse(Json& j, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        const char c = *p;
        if (c == '{') {
            j.kind = Json::Obj;
            p++;
            ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                j.obj.emplace_back(key, Json());
                if (!parse(j.obj.back().second, depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            j.kind = Json::Arr;
            p++;
            ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                j.arr.emplace_back();
                if (!parse(j.arr.back(), depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { j.kind = Json::Str; return parse_string(j.str); }
        if (!std::strncmp(p, "true", std::min<size_t>(4, end - p))) { j.kind = Json::Bool; j.b = true; p += 4; return true; }
        if (!std::strncmp(p, "false", std::min<size_t>(5, end - p))) { j.kind = Json::Bool; j.b = false; p += 5; return true; }
        if (!std::strncmp(p, "null", std::min<size_t>(4, end - p))) { j.kind = Json::Null; p += 4; return true; }
        char* e = nullptr;
        const std::string tmp(p, std::min<size_t>(64, end - p));
        const double v = std::strtod(tmp.c_str(), &e);
        if (e == tmp.c_str()) return fail("bad token");
        p += (e - tmp.c_str());
        j.kind = Json::Num;
        j.num = v;
        return true;
    }
};

bool base64_decode(const std::string& in, size_t from, std::vector<uint8_t>& out) {
    int val = 0, bits = -8;
    for (size_t i = from; i < in.size(); i++) {
        const unsigned char c = (unsigned char)in[i];
        int d;
        if (c >= 'A' && c <= 'Z') d = c - 'A';
        else if (c >= 'a' && c <= 'z') d = c - 'a' + 26;
        else if (c >= '0' && c <= '9') d = c - '0' + 52;
        else if (c == '+' || c == '-') d = 62;
        else if (c == '/' || c == '_') d = 63;
        else if (c == '=') break;
        else continue;
        val = (val << 6) | d;
        bits += 6;
        if (bits >= 0) { out.push_back((uint8_t)((val >> bits) & 0xFF)); bits -= 8; }
    }
    return true;
}

// ---------------------------------------------------------------- document access
struct Doc {
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
    std::string base_dir;
    std::string err;
};

int fail_doc(Doc& d, const std::string& m) {
    d.err = m;
    return RTHOST_ERR_FORMAT;
}

struct Accessor {
    const uint8_t* data = nullptr;
    size_t count = 0, stride = 0;
    int comp_type = 0, n_comp = 0;
    bool normalized = false;
};

size_t comp_size(int t) {
    switch (t) {
        case 5120: case 5121: return 1;
        case 5122: case 5123: return 2;
        case 5125: case 5126: return 4;
    }
    return 0;
}

bool get_accessor(Doc& d, int64_t idx, Accessor& a) {
    const Json* accs = d.root.get("accessors");
    if (!accs || idx < 0 || (size_t)idx >= accs->size()) { d.err = "accessor index out of range"; return false; }
    const Json& j = accs->arr[idx];
    if (j.has("sparse")) { d.err = "sparse accessors are not supported"; return false; }
    a.comp_type = (int)j.integer("componentType", 0);
    const int64_t count_i = j.integer("count", 0), acc_off_i = j.integer("byteOffset", 0);
    if (count_i < 0 || acc_off_i < 0) { d.err = "accessor with a negative count or byteOffset"; return false; }
    a.count = (size_t)count_i;
    a.normalized = j.get("normalized") && j.get("normalized")->b;
    const std::string type = j.string("type");
    a.n_comp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
    const size_t cs = comp_size(a.comp_type);
    if (!cs || !a.n_comp) { d.err = "unsupported accessor type"; return false; }
    const int64_t bv_idx = j.integer("bufferView", -1);
    const Json* bvs = d.root.get("bufferViews");
    if (!bvs || bv_idx < 0 || (size_t)bv_idx >= bvs->size()) { d.err = "accessor without bufferView"; return false; }
    const Json& bv = bvs->arr[bv_idx];
    const int64_t buf = bv.integer("buffer", -1);
    if (buf < 0 || (size_t)buf >= d.buffers.size()) { d.err = "bufferView.buffer out of range"; return false; }
    const int64_t bv_off_i = bv.integer("byteOffset", 0), stride_i = bv.integer("byteStride", 0);
    if (bv_off_i < 0 || stride_i < 0) { d.err = "bufferView with a negative byteOffset or byteStride"; return false; }
    const std::vector<uint8_t>& b = d.buffers[buf];
    // file-supplied numbers: every step is checked by subtraction, so that no sum or product can wrap
    if ((size_t)bv_off_i > b.size() || (size_t)acc_off_i > b.size() - (size_t)bv_off_i) { d.err = "accessor exceeds its buffer"; return false; }
    const size_t off = (size_t)bv_off_i + (size_t)acc_off_i;
    const size_t bstride = (size_t)stride_i;
    const size_t elem = cs * a.n_comp;
    a.stride = bstride ? bstride : elem;
    if (a.count) {
        const size_t room = b.size() - off;
        if (elem > room || (a.count - 1) > (room - elem) / a.stride) { d.err = "accessor exceeds its buffer"; return false; }
    }
    a.data = b.data() + off;
    return true;
}

float read_comp_f32(const Accessor& a, size_t i, int c) { // into_f32: normalised integers -> [0,1] / [-1,1]
    const uint8_t* p = a.data + i * a.stride + (size_t)c * comp_size(a.comp_type);
    switch (a.comp_type) {
        case 5126: { float v; std::memcpy(&v, p, 4); return v; }
        case 5121: return (float)p[0] / 255.0f;
        case 5123: { uint16_t v; std::memcpy(&v, p, 2); return (float)v / 65535.0f; }
        case 5120: return std::fmax((float)(int8_t)p[0] / 127.0f, -1.0f);
        case 5122: { int16_t v; std::memcpy(&v, p, 2); return std::fmax((float)v / 32767.0f, -1.0f); }
        case 5125: { uint32_t v; std::memcpy(&v, p, 4); return (float)v; }
    }
    return 0.0f;
}

uint32_t read_index(const Accessor& a, size_t i) { // into_u32
    const uint8_t* p = a.data + i * a.stride;
    switch (a.comp_type) {
        case 5121: return p[0];
        case 5123: { uint16_t v; std::memcpy(&v, p, 2); return v; }
        case 5125: { uint32_t v; std::memcpy(&v, p, 4); return v; }
    }
    return 0;
}

// gltf-crate Transform::matrix() for a decomposed node: T * R * S in f32
void trs_matrix(const float t[3], const float q[4], const float s[3], float out[16]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    const float sy2 = y2 * w, sz2 = z2 * w, sx2 = x2 * w;
    const float r[16] = {1.0f - yy2 - zz2, xy2 + sz2, xz2 - sy2, 0.0f, xy2 - sz2, 1.0f - xx2 - zz2, yz2 + sx2, 0.0f,
                         xz2 + sy2, yz2 - sx2, 1.0f - xx2 - yy2, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    const float tm[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, t[0], t[1], t[2], 1};
    const float sm[16] = {s[0], 0, 0, 0, 0, s[1], 0, 0, 0, 0, s[2], 0, 0, 0, 0, 1};
    float tr[16];
    mat4_mul(tm, r, tr);
    mat4_mul(tr, sm, out);
}

void node_matrix(const Json& node, float out[16]) {
    const Json* m = node.get("matrix");
    if (m && m->size() == 16) {
        for (int i = 0; i < 16; i++) out[i] = (float)m->arr[i].num;
        return;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    if (const Json* j = node.get("translation")) for (size_t i = 0; i < 3 && i < j->size(); i++) t[i] = (float)j->arr[i].num;
    if (const Json* j = node.get("rotation")) for (size_t i = 0; i < 4 && i < j->size(); i++) q[i] = (float)j->arr[i].num;
    if (const Json* j = node.get("scale")) for (size_t i = 0; i < 3 && i < j->size(); i++) s[i] = (float)j->arr[i].num;
    trs_matrix(t, q, s, out);
}

struct Loader {
    Doc&   d;
    Scene& s;
    std::map<int64_t, std::map<int64_t, uint32_t>> loaded_meshes; // mesh -> primitive -> scene primitive
    std::map<int64_t, TexRef> texture_map;

    int load_image_bytes(const Json& img, std::vector<uint8_t>& bytes) {
        const int64_t bv_idx = img.integer("bufferView", -1);
        if (bv_idx >= 0) {
            const Json* bvs = d.root.get("bufferViews");
            if (!bvs || (size_t)bv_idx >= bvs->size()) return fail_doc(d, "image bufferView out of range");
            const Json& bv = bvs->arr[bv_idx];
            const int64_t buf = bv.integer("buffer", -1);
            const int64_t off_i = bv.integer("byteOffset", 0), len_i = bv.integer("byteLength", 0);
            if (buf < 0 || (size_t)buf >= d.buffers.size() || off_i < 0 || len_i < 0) return fail_doc(d, "image bufferView exceeds buffer");
            const size_t off = (size_t)off_i, len = (size_t)len_i, have = d.buffers[buf].size();
            if (off > have || len > have - off) return fail_doc(d, "image bufferView exceeds buffer");
            bytes.assign(d.buffers[buf].begin() + off, d.buffers[buf].begin() + off + len);
            return RTHOST_OK;
        }
        const std::string uri = img.string("uri");
        if (uri.rfind("data:", 0) == 0) {
            const size_t comma = uri.find(',');
            if (comma == std::string::npos) return fail_doc(d, "bad data URI");
            base64_decode(uri, comma + 1, bytes);
            return RTHOST_OK;
        }
        if (!read_file(d.base_dir + "/" + uri, bytes)) { d.err = "cannot read image " + uri; return RTHOST_ERR_IO; }
        return RTHOST_OK;
    }

    // Scene::add_gltf_texture, scene.rs:299-341: one atlas entry per glTF texture index
    int add_texture(int64_t tex_idx, TexRef& out) {
        auto it = texture_map.find(tex_idx);
        if (it != texture_map.end()) { out = it->second; return RTHOST_OK; }
        const Json* texs = d.root.get("textures");
        if (!texs || tex_idx < 0 || (size_t)tex_idx >= texs->size()) return fail_doc(d, "texture index out of range");
        const int64_t src = texs->arr[tex_idx].integer("source", -1);
        const Json* imgs = d.root.get("images");
        if (!imgs || src < 0 || (size_t)src >= imgs->size()) return fail_doc(d, "texture without image source");
        std::vector<uint8_t> bytes, rgba;
        int rc = load_image_bytes(imgs->arr[src], bytes);
        if (rc != RTHOST_OK) return rc;
        uint32_t w = 0, h = 0;
        std::string e;
        // image::load_from_memory guesses the format from the content (scene.rs:310, 318)
        const bool is_jpeg = bytes.size() >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8;
        if (!(is_jpeg ? decode_jpeg(bytes.data(), bytes.size(), rgba, w, h, e) : decode_png(bytes.data(), bytes.size(), rgba, w, h, e)))
            return fail_doc(d, "image decode: " + e);
        rc = rthost::add_texture(s, rgba.data(), w, h, &out);
        if (rc != RTHOST_OK) return rc;
        texture_map[tex_idx] = out;
        return RTHOST_OK;
    }

    int texture_info(const Json* info, TexRef& ref, uint32_t& set, std::map<uint32_t, bool>& wanted_sets) {
        if (!info) return RTHOST_OK;
        const int64_t idx = info->integer("index", -1);
        int rc = add_texture(idx, ref);
        if (rc != RTHOST_OK) return rc;
        set = (uint32_t)info->integer("texCoord", 0);
        wanted_sets[set] = true; // try_load_texcoords, scene.rs:418-430
        return RTHOST_OK;
    }

    int add_primitive_from(const Json& prim, const float node_xf[16], uint32_t* out_prim) {
        if (prim.integer("mode", 4) != 4) return fail_doc(d, "non-triangle primitives are not supported"); // scene.rs:579-581
        const Json* attrs = prim.get("attributes");
        if (!attrs || !attrs->has("POSITION")) return fail_doc(d, "primitive without POSITION");
        Accessor pos;
        if (!get_accessor(d, attrs->integer("POSITION", -1), pos) || pos.comp_type != 5126 || pos.n_comp != 3) {
            if (d.err.empty()) d.err = "POSITION must be float VEC3";
            return RTHOST_ERR_FORMAT;
        }
        const size_t nv = pos.count;
        std::vector<float> positions(nv * 3);
        for (size_t i = 0; i < nv; i++) { // from_gltf_vec3: (x, y, z) -> (z, x, y), scene.rs:286-288
            positions[i * 3 + 0] = read_comp_f32(pos, i, 2);
            positions[i * 3 + 1] = read_comp_f32(pos, i, 0);
            positions[i * 3 + 2] = read_comp_f32(pos, i, 1);
        }

        // material (scene.rs:451-495); a primitive without one gets the gltf crate's defaults
        Material m;
        std::memset(&m, 0, sizeof m);
        std::map<uint32_t, bool> wanted_sets;
        const Json* mats = d.root.get("materials");
        const int64_t mat_idx = prim.integer("material", -1);
        static const Json empty_obj = [] { Json j; j.kind = Json::Obj; return j; }();
        const Json& mat = (mats && mat_idx >= 0 && (size_t)mat_idx < mats->size()) ? mats->arr[mat_idx] : empty_obj;
        const std::string alpha_mode = mat.string("alphaMode", "OPAQUE");
        m.alpha_settings = pack_alpha(alpha_mode == "MASK" ? 1u : alpha_mode == "BLEND" ? 2u : 0u, (float)mat.number("alphaCutoff", 0.5));
        const Json* pbr_p = mat.get("pbrMetallicRoughness");
        const Json& pbr = pbr_p ? *pbr_p : empty_obj;
        for (int k = 0; k < 4; k++) m.albedo_factor[k] = 1.0f;
        if (const Json* f = pbr.get("baseColorFactor")) for (size_t k = 0; k < 4 && k < f->size(); k++) m.albedo_factor[k] = (float)f->arr[k].num;
        int rc = texture_info(pbr.get("baseColorTexture"), m.albedo, m.albedo_texcoord, wanted_sets);
        if (rc != RTHOST_OK) return rc;
        m.metallic_factor = (float)pbr.number("metallicFactor", 1.0);
        m.roughness_factor = (float)pbr.number("roughnessFactor", 1.0);
        rc = texture_info(pbr.get("metallicRoughnessTexture"), m.metallic_roughness, m.metal_r_texcoord, wanted_sets);
        if (rc != RTHOST_OK) return rc;
        if (const Json* f = mat.get("emissiveFactor")) for (size_t k = 0; k < 3 && k < f->size(); k++) m.emissive_factor[k] = (float)f->arr[k].num;
        float strength = 1.0f;
        if (const Json* ext = mat.get("extensions"))
            if (const Json* es = ext->get("KHR_materials_emissive_strength")) strength = (float)es->number("emissiveStrength", 1.0);
        for (int k = 0; k < 3; k++) m.emissive_factor[k] *= strength;
        rc = texture_info(mat.get("emissiveTexture"), m.emissive, m.emissive_texcoord, wanted_sets);
        if (rc != RTHOST_OK) return rc;
        const Json* nt = mat.get("normalTexture");
        m.normal_scale = nt ? (float)nt->number("scale", 1.0) : 1.0f;
        rc = texture_info(nt, m.normal, m.normal_texcoord, wanted_sets);
        if (rc != RTHOST_OK) return rc;

        // vertex attributes (scene.rs:414-430, 497-514)
        std::vector<float> uv[2];
        for (uint32_t set = 0; set < 2; set++) {
            if (!wanted_sets.count(set)) continue;
            const std::string name = "TEXCOORD_" + std::to_string(set);
            if (!attrs->has(name.c_str())) continue;
            Accessor a;
            if (!get_accessor(d, attrs->integer(name.c_str(), -1), a) || a.n_comp != 2 || a.count != nv) {
                if (d.err.empty()) d.err = name + " must be VEC2 with one entry per vertex";
                return RTHOST_ERR_FORMAT;
            }
            uv[set].resize(nv * 2);
            for (size_t i = 0; i < nv; i++) { uv[set][i * 2] = read_comp_f32(a, i, 0); uv[set][i * 2 + 1] = read_comp_f32(a, i, 1); }
        }
        std::vector<uint32_t> colors;
        if (attrs->has("COLOR_0")) {
            Accessor a;
            if (!get_accessor(d, attrs->integer("COLOR_0", -1), a) || a.count != nv) { if (d.err.empty()) d.err = "bad COLOR_0"; return RTHOST_ERR_FORMAT; }
            colors.resize(nv);
            for (size_t i = 0; i < nv; i++) {
                uint32_t packed = 0;
                for (int c = 0; c < 4; c++) {
                    float v = c < a.n_comp ? read_comp_f32(a, i, c) : 1.0f;
                    v = std::fmin(std::fmax(v, 0.0f), 1.0f);
                    packed |= (uint32_t)std::lround(v * 255.0f) << (24 - 8 * c);
                }
                colors[i] = packed;
            }
        }
        std::vector<float> normals;
        if (attrs->has("NORMAL")) {
            Accessor a;
            if (!get_accessor(d, attrs->integer("NORMAL", -1), a) || a.n_comp != 3 || a.count != nv) { if (d.err.empty()) d.err = "bad NORMAL"; return RTHOST_ERR_FORMAT; }
            normals.resize(nv * 3);
            for (size_t i = 0; i < nv; i++) {
                // normalise in glTF axes (times the reciprocal length), then swap: scene.rs:510
                const float x = read_comp_f32(a, i, 0), y = read_comp_f32(a, i, 1), z = read_comp_f32(a, i, 2);
                const float rl = 1.0f / std::sqrt((x * x + y * y) + z * z);
                normals[i * 3 + 0] = z * rl;
                normals[i * 3 + 1] = x * rl;
                normals[i * 3 + 2] = y * rl;
            }
        }
        if (!prim.has("indices")) return fail_doc(d, "only indexed meshes are supported"); // scene.rs:547-549
        Accessor ia;
        if (!get_accessor(d, prim.integer("indices", -1), ia)) return RTHOST_ERR_FORMAT;
        std::vector<uint32_t> indices(ia.count);
        for (size_t i = 0; i < ia.count; i++) indices[i] = read_index(ia, i);

        rc = add_primitive(s, positions.data(), (uint32_t)nv, indices.data(), indices.size(), normals.empty() ? nullptr : normals.data(),
                           uv[0].empty() ? nullptr : uv[0].data(), uv[1].empty() ? nullptr : uv[1].data(),
                           colors.empty() ? nullptr : colors.data(), m, node_xf, out_prim, /*normals_are_unit=*/true);
        if (rc != RTHOST_OK) d.err = "add_primitive failed";
        return rc;
    }

    void add_camera(const Json& cam, const float gltf_xf[16]) { // Camera::from_gltf, input.rs:222-253
        const Json* persp = cam.get("perspective");
        if (!persp) return; // the reference panics on orthographic cameras; they are skipped here
        float m[16];
        from_gltf_mat4(gltf_xf, m);
        rthost_camera c;
        rthost_camera_default(&c);
        c.position[0] = m[12]; c.position[1] = m[13]; c.position[2] = m[14];
        // Mat4::to_euler(ZYX): rotation = Rz(a) * Ry(b) * Rx(c)
        const float a = std::atan2(m[1], m[0]);
        const float b = std::asin(std::fmin(std::fmax(-m[2], -1.0f), 1.0f));
        const float cc = std::atan2(m[6], m[10]);
        c.yaw = a - 3.14159265358979323846f; // "yaw is off for some reason", input.rs:229-230
        c.pitch = b;
        c.roll = cc;
        c.focus = 50.0f;
        c.fovy = (float)persp->number("yfov", 1.0);
        c.aspect = (float)persp->number("aspectRatio", 0.0);
        s.cameras.push_back(c);
    }

    int add_node(int64_t idx, const float parent[16], int depth) {
        const Json* nodes = d.root.get("nodes");
        if (!nodes || idx < 0 || (size_t)idx >= nodes->size()) return fail_doc(d, "node index out of range");
        if (depth > 256) return fail_doc(d, "node hierarchy too deep (cycle?)");
        const Json& node = nodes->arr[idx];
        float local[16], world[16], mine[16];
        node_matrix(node, local);
        mat4_mul(parent, local, world); // MatrixStack::apply, scene.rs:53-58
        from_gltf_mat4(world, mine);
        const int64_t cam = node.integer("camera", -1);
        if (cam >= 0) {
            const Json* cams = d.root.get("cameras");
            if (cams && (size_t)cam < cams->size()) add_camera(cams->arr[cam], world);
        }
        const int64_t mesh_idx = node.integer("mesh", -1);
        if (mesh_idx >= 0) {
            const Json* meshes = d.root.get("meshes");
            if (!meshes || (size_t)mesh_idx >= meshes->size()) return fail_doc(d, "mesh index out of range");
            const Json* prims = meshes->arr[mesh_idx].get("primitives");
            const size_t np = prims ? prims->size() : 0;
            auto it = loaded_meshes.find(mesh_idx);
            if (it != loaded_meshes.end()) { // instancing, scene.rs:375-386
                for (size_t p = 0; p < np; p++) {
                    auto pit = it->second.find((int64_t)p);
                    if (pit != it->second.end()) {
                        int rc = add_instance(s, pit->second, mine, nullptr);
                        if (rc != RTHOST_OK) return rc;
                    }
                }
            } else {
                std::map<int64_t, uint32_t> loaded;
                for (size_t p = 0; p < np; p++) {
                    uint32_t prim_idx = 0;
                    int rc = add_primitive_from(prims->arr[p], mine, &prim_idx);
                    if (rc != RTHOST_OK) return rc;
                    loaded[(int64_t)p] = prim_idx;
                }
                loaded_meshes[mesh_idx] = loaded;
            }
        }
        if (const Json* ch = node.get("children"))
            for (const Json& c : ch->arr) {
                int rc = add_node((int64_t)c.num, world, depth + 1);
                if (rc != RTHOST_OK) return rc;
            }
        return RTHOST_OK;
    }
};

} // namespace

int load_gltf(Scene& s, const uint8_t* bytes, size_t n, const std::string& base_dir) {
    Doc d;
    d.base_dir = base_dir;
    std::vector<uint8_t> glb_bin;
    const char* json_begin = reinterpret_cast<const char*>(bytes);
    size_t json_len = n;
    bool have_bin = false;
    if (n >= 12 && !std::memcmp(bytes, "glTF", 4)) { // GLB container
        uint32_t total;
        std::memcpy(&total, bytes + 8, 4);
        size_t pos = 12;
        json_len = 0;
        while (pos + 8 <= n && pos + 8 <= total) {
            uint32_t len, kind;
            std::memcpy(&len, bytes + pos, 4);
            std::memcpy(&kind, bytes + pos + 4, 4);
            if (pos + 8 + (size_t)len > n) { set_error("gltf: truncated GLB chunk"); return RTHOST_ERR_FORMAT; }
            if (kind == 0x4E4F534Au) { json_begin = reinterpret_cast<const char*>(bytes + pos + 8); json_len = len; }
            else if (kind == 0x004E4942u && !have_bin) { glb_bin.assign(bytes + pos + 8, bytes + pos + 8 + len); have_bin = true; }
            pos += 8 + (size_t)len;
        }
        if (!json_len) { set_error("gltf: GLB without JSON chunk"); return RTHOST_ERR_FORMAT; }
    }
    JsonParser jp{json_begin, json_begin + json_len, ""};
    if (!jp.parse(d.root)) { set_error("gltf: JSON: " + jp.err); return RTHOST_ERR_FORMAT; }

    if (const Json* bufs = d.root.get("buffers")) {
        for (const Json& b : bufs->arr) {
            const std::string uri = b.string("uri");
            std::vector<uint8_t> data;
            if (uri.empty()) {
                if (!have_bin) { set_error("gltf: buffer without uri and no GLB BIN chunk"); return RTHOST_ERR_FORMAT; }
                data = glb_bin;
            } else if (uri.rfind("data:", 0) == 0) {
                const size_t comma = uri.find(',');
                if (comma == std::string::npos) { set_error("gltf: bad data URI"); return RTHOST_ERR_FORMAT; }
                base64_decode(uri, comma + 1, data);
            } else if (!read_file(base_dir + "/" + uri, data)) {
                set_error("gltf: cannot read buffer " + uri);
                return RTHOST_ERR_IO;
            }
            d.buffers.push_back(std::move(data));
        }
    }

    Loader L{d, s, {}, {}};
    static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (const Json* scenes = d.root.get("scenes")) { // every scene of the document, scene.rs:278-282
        for (const Json& sc : scenes->arr)
            if (const Json* roots = sc.get("nodes"))
                for (const Json& r : roots->arr) {
                    int rc = L.add_node((int64_t)r.num, identity, 0);
                    if (rc != RTHOST_OK) { set_error("gltf: " + d.err); return rc; }
                }
    }
    return RTHOST_OK;
}

} // namespace rthost
