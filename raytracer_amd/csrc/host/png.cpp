// PNG -> rgba8 rows (top row first), the form add_gltf_texture keeps after image.to_rgba8()
// (scene.rs:310-331).  Non-interlaced images, colour types 0/2/3/4/6, bit depths 1-16; inflate
// through zlib.  16-bit samples are scaled to 8 bits with rounding.
#include "rthost_internal.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>

namespace rthost {

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

} // namespace

bool decode_png(const uint8_t* b, size_t n, std::vector<uint8_t>& out, uint32_t& w, uint32_t& h, std::string& err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || std::memcmp(b, sig, 8) != 0) { err = "not a PNG"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    bool have_ihdr = false;
    std::vector<uint8_t> idat, palette, trns;
    while (pos + 12 <= n) {
        const uint32_t len = be32(b + pos);
        const uint8_t* type = b + pos + 4;
        const uint8_t* data = b + pos + 8;
        if (pos + 12 + (size_t)len > n) { err = "truncated chunk"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0) { err = "missing IHDR"; return false; }
    if (w > 65535u || h > 65535u) { err = "image larger than 65535 pixels a side (texture sizes are packed in 16 bits, scene.rs:82)"; return false; }
    if (interlace != 0) { err = "interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "bad colour type"; return false;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) { err = "bad bit depth"; return false; }
    if (ctype == 3 && palette.empty()) { err = "palette image without PLTE"; return false; }
    const size_t bpp_bits = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) { err = "inflate failed"; return false; }

    // undo the per-row filters in place
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* row = &raw[(stride + 1) * (size_t)y];
        const uint8_t ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, up = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
            int v = cur[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += up; break;
                case 3: v += (a + up) / 2; break;
                case 4: v += paeth(a, up, c); break;
                default: err = "bad filter type"; return false;
            }
            cur[x] = (uint8_t)v;
        }
        std::memcpy(prev.data(), cur, stride);
    }

    out.assign((size_t)w * h * 4, 255);
    const uint32_t maxv = (1u << depth) - 1u;
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* cur = &raw[(stride + 1) * (size_t)y + 1];
        uint8_t* o = &out[(size_t)y * w * 4];
        for (uint32_t x = 0; x < w; x++) {
            uint32_t s[4] = {0, 0, 0, 0};
            for (int c = 0; c < channels; c++) {
                const size_t bit = ((size_t)x * channels + c) * depth;
                if (depth == 8) s[c] = cur[bit / 8];
                else if (depth == 16) s[c] = ((uint32_t)cur[bit / 8] << 8) | cur[bit / 8 + 1];
                else s[c] = (cur[bit / 8] >> (8 - depth - (bit % 8))) & maxv;
            }
            auto to8 = [&](uint32_t v) -> uint8_t { return depth == 8 ? (uint8_t)v : (uint8_t)((v * 255u + maxv / 2u) / maxv); };
            uint8_t r, g, bl, a = 255;
            if (ctype == 3) {
                const uint32_t i = s[0];
                if ((size_t)i * 3 + 2 >= palette.size()) { err = "palette index out of range"; return false; }
                r = palette[i * 3]; g = palette[i * 3 + 1]; bl = palette[i * 3 + 2];
                if (i < trns.size()) a = trns[i];
            } else if (ctype == 0 || ctype == 4) {
                r = g = bl = to8(s[0]);
                if (ctype == 4) a = to8(s[1]);
                else if (trns.size() >= 2 && s[0] == (((uint32_t)trns[0] << 8) | trns[1])) a = 0;
            } else {
                r = to8(s[0]); g = to8(s[1]); bl = to8(s[2]);
                if (ctype == 6) a = to8(s[3]);
                else if (trns.size() >= 6 && s[0] == (((uint32_t)trns[0] << 8) | trns[1]) && s[1] == (((uint32_t)trns[2] << 8) | trns[3]) &&
                         s[2] == (((uint32_t)trns[4] << 8) | trns[5])) a = 0;
            }
            o[x * 4] = r; o[x * 4 + 1] = g; o[x * 4 + 2] = bl; o[x * 4 + 3] = a;
        }
    }
    return true;
}

} // namespace rthost
