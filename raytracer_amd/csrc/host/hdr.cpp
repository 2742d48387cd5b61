// Radiance .hdr (RGBE, flat or new-style RLE scanlines) -> rgba32f rows, top row first: what
// set_equirectangular_env_map (scene.rs:603-615) stores after image::load_from_memory +
// into_rgba32f.  A pixel converts as mantissa * 2^(exponent - 136), zero when exponent == 0;
// alpha is 1.
#include "rthost_internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace rthost {

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == out.size();
}

namespace {

bool read_line(const uint8_t* b, size_t n, size_t& pos, std::string& line) {
    line.clear();
    while (pos < n) {
        const char c = (char)b[pos++];
        if (c == '\n') return true;
        line.push_back(c);
    }
    return !line.empty();
}

} // namespace

bool decode_hdr(const uint8_t* b, size_t n, std::vector<float>& out, uint32_t& w, uint32_t& h, std::string& err) {
    size_t pos = 0;
    std::string line;
    if (!read_line(b, n, pos, line) || (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0)) {
        err = "not a Radiance file";
        return false;
    }
    bool format_ok = false;
    for (;;) {
        if (!read_line(b, n, pos, line)) { err = "truncated header"; return false; }
        if (line.empty()) break;
        if (line.rfind("FORMAT=", 0) == 0) format_ok = line.find("32-bit_rle_rgbe") != std::string::npos;
    }
    if (!format_ok) { err = "FORMAT is not 32-bit_rle_rgbe"; return false; }
    if (!read_line(b, n, pos, line)) { err = "missing resolution line"; return false; }
    int hh = 0, ww = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &hh, &ww) != 2 || hh <= 0 || ww <= 0) {
        err = "unsupported orientation '" + line + "' (only -Y h +X w)";
        return false;
    }
    // before anything is allocated: sane dimensions, and at least the 4 bytes a scanline cannot do without
    if (ww > 65536 || hh > 65536 || (size_t)hh * 4 > n - pos) { err = "resolution line does not fit the file"; return false; }
    w = (uint32_t)ww;
    h = (uint32_t)hh;
    out.assign((size_t)w * h * 4, 1.0f);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (uint32_t y = 0; y < h; y++) {
        if (pos + 4 > n) { err = "truncated pixel data"; return false; }
        const bool rle = w >= 8 && w < 32768 && b[pos] == 2 && b[pos + 1] == 2 && ((b[pos + 2] << 8) | b[pos + 3]) == (int)w;
        if (rle) {
            pos += 4;
            for (int ch = 0; ch < 4; ch++) {
                uint32_t x = 0;
                while (x < w) {
                    if (pos >= n) { err = "truncated RLE data"; return false; }
                    uint8_t count = b[pos++];
                    if (count > 128) {
                        count -= 128;
                        if (pos >= n || x + count > w) { err = "bad RLE run"; return false; }
                        const uint8_t v = b[pos++];
                        for (uint8_t k = 0; k < count; k++) scan[(size_t)(x++) * 4 + ch] = v;
                    } else {
                        if (count == 0 || pos + count > n || x + count > w) { err = "bad RLE literal"; return false; }
                        for (uint8_t k = 0; k < count; k++) scan[(size_t)(x++) * 4 + ch] = b[pos++];
                    }
                }
            }
        } else {
            if (pos + (size_t)w * 4 > n) { err = "truncated flat data"; return false; }
            std::memcpy(scan.data(), b + pos, (size_t)w * 4);
            pos += (size_t)w * 4;
        }
        float* row = &out[(size_t)y * w * 4];
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t* p = &scan[(size_t)x * 4];
            if (p[3] == 0) {
                row[x * 4] = row[x * 4 + 1] = row[x * 4 + 2] = 0.0f;
            } else {
                const float e = std::exp2((float)p[3] - (128.0f + 8.0f));
                row[x * 4] = e * (float)p[0];
                row[x * 4 + 1] = e * (float)p[1];
                row[x * 4 + 2] = e * (float)p[2];
            }
            row[x * 4 + 3] = 1.0f;
        }
    }
    return true;
}

} // namespace rthost
