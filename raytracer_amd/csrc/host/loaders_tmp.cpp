// temporary: replaced by gltf.cpp / png.cpp / hdr.cpp
#include "rthost_internal.h"
#include <cstdio>
namespace rthost {
bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == out.size();
}
int load_gltf(Scene&, const uint8_t*, size_t, const std::string&) { set_error("gltf loader not built"); return RTHOST_ERR_FORMAT; }
bool decode_png(const uint8_t*, size_t, std::vector<uint8_t>&, uint32_t&, uint32_t&, std::string& e) { e = "not built"; return false; }
bool decode_hdr(const uint8_t*, size_t, std::vector<float>&, uint32_t&, uint32_t&, std::string& e) { e = "not built"; return false; }
}
