// JPEG (ITU T.81) -> rgba8 for glTF textures; the reference gets its texels from the image crate
// (image::load_from_memory(..).to_rgba8(), scene.rs:310-322), whose JPEG decoder follows
// libjpeg's arithmetic.  Supported: 8-bit baseline and progressive Huffman, grayscale and
// YCbCr (or Adobe RGB), sampling factors 1-2 per axis with libjpeg's "fancy" triangle
// upsampling (other ratios are replicated), restart intervals.  Not supported: arithmetic
// coding, lossless, 12-bit, CMYK.  The inverse DCT is libjpeg's accurate integer one
// (13-bit constants, two passes) and the colour conversion uses its 16-bit fixed-point
// coefficients, so that texels agree with that family of decoders.
#include "rthost_internal.h"

#include <algorithm>
#include <cstring>

namespace rthost {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool     present = false;
    uint8_t  lookup_len[512];   // code length for the next 9 bits, 0 = longer than 9
    uint8_t  lookup_sym[512];
    int32_t  maxcode[18];       // largest code of each length, -1 if none
    int32_t  valptr[17];
    int32_t  mincode[17];
    uint8_t  vals[256];

    bool build(const uint8_t counts[16], const uint8_t* symbols, int n_symbols) {
        std::memset(lookup_len, 0, sizeof lookup_len);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
                if (k >= n_symbols || code >= (1 << len)) return false;
                vals[k] = symbols[k];
                if (len <= 9) {
                    const int first = code << (9 - len), n = 1 << (9 - len);
                    for (int j = 0; j < n; j++) { lookup_len[first + j] = (uint8_t)len; lookup_sym[first + j] = symbols[k]; }
                }
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;            // tables of the current scan
    int w = 0, h_px = 0;           // true size in samples: ceil(image * h / hmax)
    int blocks_w = 0, blocks_h = 0;   // blocks covering w x h_px (non-interleaved scans)
    int stride_blocks = 0, rows_blocks = 0; // allocated: whole MCUs
    std::vector<int16_t> coef;     // stride_blocks * rows_blocks * 64, natural order
    std::vector<uint8_t> plane;    // (stride_blocks * 8) x (rows_blocks * 8)
    int pred = 0;
};

struct Decoder {
    const uint8_t* p;
    const uint8_t* end;
    std::string    err;
    uint16_t qt[4][64];
    bool     qt_present[4] = {false, false, false, false};
    Huff     dc[4], ac[4];
    std::vector<Component> comps;
    int  width = 0, height = 0, hmax = 1, vmax = 1;
    int  mcus_x = 0, mcus_y = 0;
    bool progressive = false, have_frame = false;
    int  restart_interval = 0;
    int  adobe_transform = -1;
    // entropy-coded segment reader
    uint32_t bitbuf = 0;
    int      bitcnt = 0;
    bool     hit_marker = false;
    uint32_t eobrun = 0;

    bool fail(const char* m) { if (err.empty()) err = m; return false; }

    void reset_bits() { bitbuf = 0; bitcnt = 0; hit_marker = false; }
    void fill() {
        while (bitcnt <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) { p += 2; }
                    else { hit_marker = true; byte = 0; } // a marker: feed zeros, leave p on it
                } else {
                    p++;
                }
            } else {
                hit_marker = true;
            }
            bitbuf |= byte << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) fill();
        const int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n;
        bitcnt -= n;
        return v;
    }
    int bit() { return bits(1); }
    static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int receive_extend(int n) { return n ? extend(bits(n), n) : 0; }
    int decode(const Huff& h) {
        if (bitcnt < 16) fill();
        const int look = (int)(bitbuf >> 23);
        int len = h.lookup_len[look];
        if (len) {
            bitbuf <<= len;
            bitcnt -= len;
            return h.lookup_sym[look];
        }
        int code = (int)(bitbuf >> 22); // 10 bits
        len = 10;
        while (len <= 16 && code > h.maxcode[len]) {
            len++;
            code = (int)(bitbuf >> (32 - len));
        }
        if (len > 16) { fail("bad Huffman code"); return 0; }
        bitbuf <<= len;
        bitcnt -= len;
        return h.vals[(h.valptr[len] + code - h.mincode[len]) & 255];
    }

    // ---------------------------------------------------------------- markers
    static int be16(const uint8_t* q) { return (q[0] << 8) | q[1]; }

    bool read_dqt(const uint8_t* q, int len) {
        while (len > 0) {
            const int pq = q[0] >> 4, tq = q[0] & 15;
            if (tq > 3 || pq > 1) return fail("bad DQT");
            const int need = 1 + 64 * (pq + 1);
            if (len < need) return fail("short DQT");
            for (int i = 0; i < 64; i++) qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(q + 1 + 2 * i) : q[1 + i];
            qt_present[tq] = true;
            q += need;
            len -= need;
        }
        return true;
    }
    bool read_dht(const uint8_t* q, int len) {
        while (len > 0) {
            if (len < 17) return fail("short DHT");
            const int tc = q[0] >> 4, th = q[0] & 15;
            if (tc > 1 || th > 3) return fail("bad DHT");
            int n = 0;
            for (int i = 0; i < 16; i++) n += q[1 + i];
            if (n > 256 || len < 17 + n) return fail("bad DHT length");
            Huff& h = tc ? ac[th] : dc[th];
            if (!h.build(q + 1, q + 17, n)) return fail("bad DHT codes");
            q += 17 + n;
            len -= 17 + n;
        }
        return true;
    }
    bool read_sof(const uint8_t* q, int len, bool prog) {
        if (have_frame) return fail("second frame header");
        if (len < 6) return fail("short SOF");
        if (q[0] != 8) return fail("only 8-bit JPEG is supported");
        height = be16(q + 1);
        width = be16(q + 3);
        const int nc = q[5];
        if (width == 0 || height == 0) return fail("empty JPEG image");
        if (nc != 1 && nc != 3) return fail("only grayscale and three-component JPEG are supported");
        if (len < 6 + 3 * nc) return fail("short SOF");
        comps.resize(nc);
        for (int i = 0; i < nc; i++) {
            Component& c = comps[i];
            c.id = q[6 + 3 * i];
            c.h = q[7 + 3 * i] >> 4;
            c.v = q[7 + 3 * i] & 15;
            c.tq = q[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component in SOF");
            hmax = std::max(hmax, c.h);
            vmax = std::max(vmax, c.v);
        }
        if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; } // a single component is never interleaved
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
        mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component& c : comps) {
            c.w = (width * c.h + hmax - 1) / hmax;
            c.h_px = (height * c.v + vmax - 1) / vmax;
            c.blocks_w = (c.w + 7) / 8;
            c.blocks_h = (c.h_px + 7) / 8;
            c.stride_blocks = mcus_x * c.h;
            c.rows_blocks = mcus_y * c.v;
            const size_t nb = (size_t)c.stride_blocks * c.rows_blocks;
            if (nb * 64 > ((size_t)1 << 31)) return fail("JPEG image too large");
            c.coef.assign(nb * 64, 0);
        }
        progressive = prog;
        have_frame = true;
        return true;
    }

    // ---------------------------------------------------------------- block decoding
    bool baseline_block(Component& c, int16_t* blk) {
        const int t = decode(dc[c.td]);
        if (t > 11) return fail("bad DC size");
        c.pred += receive_extend(t);
        blk[0] = (int16_t)c.pred;
        const Huff& h = ac[c.ta];
        for (int k = 1; k < 64;) {
            const int rs = decode(h), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) return fail("AC run beyond block");
                blk[kZigzag[k++]] = (int16_t)receive_extend(s);
            }
        }
        return true;
    }
    bool prog_dc(Component& c, int16_t* blk, int ah, int al) {
        if (ah == 0) {
            const int t = decode(dc[c.td]);
            if (t > 11) return fail("bad DC size");
            c.pred += receive_extend(t);
            blk[0] = (int16_t)(c.pred * (1 << al));
        } else if (bit()) {
            blk[0] = (int16_t)(blk[0] | (1 << al));
        }
        return true;
    }
    bool prog_ac_first(Component& c, int16_t* blk, int ss, int se, int al) {
        if (eobrun) { eobrun--; return true; }
        const Huff& h = ac[c.ta];
        for (int k = ss; k <= se;) {
            const int rs = decode(h), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1u << r) - 1u;
                    if (r) eobrun += (uint32_t)bits(r);
                    break;
                }
                k += 16;
            } else {
                k += r;
                if (k > 63) return fail("AC run beyond block");
                blk[kZigzag[k++]] = (int16_t)(receive_extend(s) * (1 << al));
            }
        }
        return true;
    }
    bool prog_ac_refine(Component& c, int16_t* blk, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        const Huff& h = ac[c.ta];
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se;) {
                const int rs = decode(h);
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1u << r);
                        if (r) eobrun += (uint32_t)bits(r);
                        break; // the rest of the band is handled below as part of the run
                    }
                } else {
                    if (s != 1) return fail("bad refinement code");
                    value = bit() ? p1 : m1;
                }
                // skip r zero-history coefficients, correcting the non-zero ones on the way
                while (k <= se) {
                    int16_t& cf = blk[kZigzag[k]];
                    if (cf != 0) {
                        if (bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                    } else {
                        if (r == 0) {
                            if (value) cf = (int16_t)value;
                            k++;
                            break;
                        }
                        r--;
                    }
                    k++;
                }
            }
        }
        if (eobrun) {
            for (; k <= se; k++) {
                int16_t& cf = blk[kZigzag[k]];
                if (cf != 0 && bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
            }
            eobrun--;
        }
        return true;
    }

    bool restart() {
        // p sits on (or before) the RSTn marker: skip to it
        reset_bits();
        while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
            if (p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF) return true; // another marker: let the caller see it
            p++;
        }
        if (p + 1 < end) p += 2;
        for (Component& c : comps) c.pred = 0;
        eobrun = 0;
        return true;
    }

    bool read_scan(const uint8_t* q, int len) {
        if (!have_frame) return fail("scan before frame header");
        if (len < 1) return fail("short SOS");
        const int ns = q[0];
        if (ns < 1 || ns > (int)comps.size() || len < 1 + 2 * ns + 3) return fail("bad SOS");
        Component* sc[3];
        for (int i = 0; i < ns; i++) {
            sc[i] = nullptr;
            for (Component& c : comps)
                if (c.id == q[1 + 2 * i]) sc[i] = &c;
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->td = q[2 + 2 * i] >> 4;
            sc[i]->ta = q[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table index in SOS");
        }
        const int ss = q[1 + 2 * ns], se = q[2 + 2 * ns], ah = q[3 + 2 * ns] >> 4, al = q[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss != 0 && ns != 1) || al > 13) return fail("bad progressive scan parameters");
        }
        for (int i = 0; i < ns; i++) {
            const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss != 0;
            if (need_dc && !(progressive && ah != 0) && !dc[sc[i]->td].present) return fail("missing DC Huffman table");
            if (need_ac && !ac[sc[i]->ta].present) return fail("missing AC Huffman table");
        }
        reset_bits();
        for (Component& c : comps) c.pred = 0;
        eobrun = 0;

        auto one_block = [&](Component& c, int bx, int by) -> bool {
            int16_t* blk = &c.coef[((size_t)by * c.stride_blocks + bx) * 64];
            if (!progressive) return baseline_block(c, blk);
            if (ss == 0) return prog_dc(c, blk, ah, al);
            return ah == 0 ? prog_ac_first(c, blk, ss, se, al) : prog_ac_refine(c, blk, ss, se, al);
        };
        int until_restart = restart_interval;
        auto count_mcu = [&]() -> bool {
            if (restart_interval && --until_restart == 0) {
                until_restart = restart_interval;
                return restart();
            }
            return true;
        };
        if (ns == 1) { // non-interleaved: the component's own blocks, row by row
            Component& c = *sc[0];
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++) {
                    if (!one_block(c, bx, by)) return false;
                    const bool last = by == c.blocks_h - 1 && bx == c.blocks_w - 1;
                    if (!last && !count_mcu()) return false;
                }
        } else {
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    for (int i = 0; i < ns; i++) {
                        Component& c = *sc[i];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++)
                                if (!one_block(c, mx * c.h + h, my * c.v + v)) return false;
                    }
                    const bool last = my == mcus_y - 1 && mx == mcus_x - 1;
                    if (!last && !count_mcu()) return false;
                }
        }
        if (!err.empty()) return false;
        // leave p at the next marker
        while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
        return true;
    }

    // ---------------------------------------------------------------- inverse DCT (libjpeg jidctint)
    static inline int descale(int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); }
    static void idct(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
        const int CB = 13, P1 = 2;
        int ws[64];
        for (int col = 0; col < 8; col++) {
            const int64_t i0 = in[col] * q[col], i1 = in[8 + col] * q[8 + col], i2 = in[16 + col] * q[16 + col],
                          i3 = in[24 + col] * q[24 + col], i4 = in[32 + col] * q[32 + col], i5 = in[40 + col] * q[40 + col],
                          i6 = in[48 + col] * q[48 + col], i7 = in[56 + col] * q[56 + col];
            int64_t z1 = (i2 + i6) * 4433;
            const int64_t t2 = z1 + i6 * -15137, t3 = z1 + i2 * 6270;
            const int64_t t0 = (i0 + i4) * (1 << CB), t1 = (i0 - i4) * (1 << CB);
            const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            int64_t o0 = i7, o1 = i5, o2 = i3, o3 = i1;
            z1 = o0 + o3;
            int64_t z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
            const int64_t z5 = (z3 + z4) * 9633;
            o0 *= 2446; o1 *= 16819; o2 *= 25172; o3 *= 12299;
            z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
            z3 += z5; z4 += z5;
            o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
            ws[col] = descale(t10 + o3, CB - P1);      ws[56 + col] = descale(t10 - o3, CB - P1);
            ws[8 + col] = descale(t11 + o2, CB - P1);  ws[48 + col] = descale(t11 - o2, CB - P1);
            ws[16 + col] = descale(t12 + o1, CB - P1); ws[40 + col] = descale(t12 - o1, CB - P1);
            ws[24 + col] = descale(t13 + o0, CB - P1); ws[32 + col] = descale(t13 - o0, CB - P1);
        }
        for (int row = 0; row < 8; row++) {
            const int* w = ws + row * 8;
            const int64_t i0 = w[0], i1 = w[1], i2 = w[2], i3 = w[3], i4 = w[4], i5 = w[5], i6 = w[6], i7 = w[7];
            int64_t z1 = (i2 + i6) * 4433;
            const int64_t t2 = z1 + i6 * -15137, t3 = z1 + i2 * 6270;
            const int64_t t0 = (i0 + i4) * (1 << CB), t1 = (i0 - i4) * (1 << CB);
            const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            int64_t o0 = i7, o1 = i5, o2 = i3, o3 = i1;
            z1 = o0 + o3;
            int64_t z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
            const int64_t z5 = (z3 + z4) * 9633;
            o0 *= 2446; o1 *= 16819; o2 *= 25172; o3 *= 12299;
            z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
            z3 += z5; z4 += z5;
            o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
            const int S = CB + P1 + 3;
            const int v[8] = {descale(t10 + o3, S), descale(t11 + o2, S), descale(t12 + o1, S), descale(t13 + o0, S),
                              descale(t13 - o0, S), descale(t12 - o1, S), descale(t11 - o2, S), descale(t10 - o3, S)};
            uint8_t* o = out + (size_t)row * stride;
            for (int k = 0; k < 8; k++) o[k] = (uint8_t)std::min(255, std::max(0, v[k] + 128));
        }
    }

    // ---------------------------------------------------------------- upsampling (libjpeg jdsample)
    // Brings component c to full resolution: full is width x height, row-major.
    void upsample(const Component& c, std::vector<uint8_t>& full) const {
        const int ps = c.stride_blocks * 8;
        full.resize((size_t)width * height);
        const int fx = hmax / c.h, fy = vmax / c.v;
        // the triangle filters need three samples in a row; narrower components are replicated
        const bool exact = c.h * fx == hmax && c.v * fy == vmax && (fx == 1 || c.w > 2);
        auto src = [&](int x, int y) -> int {
            x = std::min(std::max(x, 0), c.w - 1);
            y = std::min(std::max(y, 0), c.h_px - 1);
            return c.plane[(size_t)y * ps + x];
        };
        if (exact && fx == 1 && fy == 1) {
            for (int y = 0; y < height; y++) std::memcpy(&full[(size_t)y * width], &c.plane[(size_t)y * ps], width);
        } else if (exact && fx == 2 && fy == 1) { // h2v1 fancy: 3/4 nearer + 1/4 farther
            for (int y = 0; y < height; y++) {
                uint8_t* o = &full[(size_t)y * width];
                for (int x = 0; x < width; x++) {
                    const int i = x >> 1, s = src(i, y);
                    int v;
                    if (c.w == 1) v = s;
                    else if ((x & 1) == 0) v = i == 0 ? s : (3 * s + src(i - 1, y) + 1) >> 2;
                    else v = i == c.w - 1 ? s : (3 * s + src(i + 1, y) + 2) >> 2;
                    o[x] = (uint8_t)v;
                }
            }
        } else if (exact && fx == 1 && fy == 2) { // h1v2 fancy
            for (int y = 0; y < height; y++) {
                const int j = y >> 1, near_row = j, far_row = (y & 1) ? j + 1 : j - 1, bias = (y & 1) ? 2 : 1;
                uint8_t* o = &full[(size_t)y * width];
                for (int x = 0; x < width; x++) o[x] = (uint8_t)((3 * src(x, near_row) + src(x, far_row) + bias) >> 2);
            }
        } else if (exact && fx == 2 && fy == 2) { // h2v2 fancy: 9/16, 3/16, 3/16, 1/16
            std::vector<int> colsum(c.w);
            for (int y = 0; y < height; y++) {
                const int j = y >> 1, far_row = (y & 1) ? j + 1 : j - 1;
                for (int i = 0; i < c.w; i++) colsum[i] = 3 * src(i, j) + src(i, far_row);
                uint8_t* o = &full[(size_t)y * width];
                for (int x = 0; x < width; x++) {
                    const int i = x >> 1, t = colsum[i];
                    int v;
                    if (c.w == 1) v = (x & 1) == 0 ? (t * 4 + 8) >> 4 : (t * 4 + 7) >> 4;
                    else if ((x & 1) == 0) v = i == 0 ? (t * 4 + 8) >> 4 : (t * 3 + colsum[i - 1] + 8) >> 4;
                    else v = i == c.w - 1 ? (t * 4 + 7) >> 4 : (t * 3 + colsum[i + 1] + 7) >> 4;
                    o[x] = (uint8_t)v;
                }
            }
        } else { // any other ratio: nearest sample
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) full[(size_t)y * width + x] = (uint8_t)src(x * c.h / hmax, y * c.v / vmax);
        }
    }

    bool finish(std::vector<uint8_t>& rgba) {
        for (Component& c : comps) {
            if (!qt_present[c.tq]) return fail("missing quantisation table");
            const int ps = c.stride_blocks * 8;
            c.plane.assign((size_t)ps * c.rows_blocks * 8, 0);
            for (int by = 0; by < c.rows_blocks; by++)
                for (int bx = 0; bx < c.stride_blocks; bx++)
                    idct(&c.coef[((size_t)by * c.stride_blocks + bx) * 64], qt[c.tq], &c.plane[(size_t)by * 8 * ps + bx * 8], ps);
            c.coef.clear();
            c.coef.shrink_to_fit();
        }
        rgba.resize((size_t)width * height * 4);
        std::vector<uint8_t> f0, f1, f2;
        upsample(comps[0], f0);
        if (comps.size() == 1) {
            for (size_t i = 0; i < f0.size(); i++) { rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = f0[i]; rgba[4 * i + 3] = 255; }
            return true;
        }
        upsample(comps[1], f1);
        upsample(comps[2], f2);
        const bool is_rgb = adobe_transform == 0 || (adobe_transform < 0 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
        auto clamp8 = [](int v) { return (uint8_t)std::min(255, std::max(0, v)); };
        for (size_t i = 0; i < f0.size(); i++) {
            int r = f0[i], g = f1[i], b = f2[i];
            if (!is_rgb) { // libjpeg jdcolor: 16-bit fixed point
                const int y = f0[i], cb = f1[i] - 128, cr = f2[i] - 128;
                r = y + ((91881 * cr + 32768) >> 16);
                g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                b = y + ((116130 * cb + 32768) >> 16);
            }
            rgba[4 * i] = clamp8(r);
            rgba[4 * i + 1] = clamp8(g);
            rgba[4 * i + 2] = clamp8(b);
            rgba[4 * i + 3] = 255;
        }
        return true;
    }

    bool run(std::vector<uint8_t>& rgba) {
        if (end - p < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG stream");
        p += 2;
        bool seen_scan = false;
        for (;;) {
            while (p < end && *p != 0xFF) p++;      // garbage before a marker
            while (p < end && *p == 0xFF) p++;      // fill bytes
            if (p >= end) break;
            const int m = *p++;
            if (m == 0xD9) break;                   // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (end - p < 2) return fail("truncated JPEG marker");
            const int len = be16(p);
            if (len < 2 || end - p < len) return fail("truncated JPEG segment");
            const uint8_t* q = p + 2;
            const int n = len - 2;
            p += len;
            bool ok = true;
            switch (m) {
                case 0xDB: ok = read_dqt(q, n); break;
                case 0xC4: ok = read_dht(q, n); break;
                case 0xC0: case 0xC1: ok = read_sof(q, n, false); break;
                case 0xC2: ok = read_sof(q, n, true); break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return fail("JPEG coding process not supported (lossless, hierarchical or arithmetic)");
                case 0xDD: if (n < 2) return fail("short DRI"); restart_interval = be16(q); break;
                case 0xEE: if (n >= 12 && std::memcmp(q, "Adobe", 5) == 0) adobe_transform = q[11]; break;
                case 0xDA: ok = read_scan(q, n); seen_scan = true; break;
                default: break; // APPn, COM, DNL...: skipped
            }
            if (!ok) return false;
        }
        if (!have_frame || !seen_scan) return fail("JPEG stream without image data");
        return finish(rgba);
    }
};

} // namespace

bool decode_jpeg(const uint8_t* bytes, size_t n, std::vector<uint8_t>& rgba8, uint32_t& w, uint32_t& h, std::string& err) {
    Decoder d;
    d.p = bytes;
    d.end = bytes + n;
    if (!d.run(rgba8)) {
        err = "JPEG: " + (d.err.empty() ? std::string("decode failed") : d.err);
        return false;
    }
    w = (uint32_t)d.width;
    h = (uint32_t)d.height;
    return true;
}

} // namespace rthost
