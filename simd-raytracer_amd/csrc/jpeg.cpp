// Baseline JPEG decoder for bitmap textures (scene/texture/bitmap.hpp:11-37 calls `stbi_load(path, &w, &h, &channels, 0)`).
//
// stb_image is not part of /root/reference (CMakeLists.txt:17-21 fetches github.com/nothings/stb `master` at configure time) and
// not in this image, so this file restates the published algorithm of stb_image.h (v2.2x - v2.30, `stbi__jpeg_*`) for what the
// reference's scenes use: sequential baseline JPEG (SOF0), 8 bit, Huffman coded, 1 or 3 components.  The texel bytes decide
// pixel values, so the arithmetic is stb's, not libjpeg's:
//   - stbi__jpeg_decode_block: coefficient = (short)(value * dequant[zig])
//   - stbi__idct_block: 12-bit fixed point, stbi__f2f(x) = (int)(x * 4096 + 0.5); columns keep 2 extra bits ((x + 512) >> 10) and
//     short-cut an all-zero AC column to d[0] * 4; rows (x + 65536 + (128 << 17)) >> 17, clamped to a byte
//   - chroma expansion: stbi__resample_row_{generic,v_2,h_2,hv_2}
//   - stbi__YCbCr_to_RGB_row: 20-bit fixed point, stbi__float2fixed(x) = ((int)(x * 4096.0f + 0.5f)) << 8, and the Cb term of
//     green masked with 0xffff0000
// Pinned by the reference's own render outputs/textures.png of scenes/hw12/scene4 (tests/test_reference_outputs.py): every pixel
// of the textured quad equals, which a libjpeg decode of the same file does not achieve (1,959 pixels differ).
// Progressive / arithmetic-coded / 12-bit / CMYK files and other containers (PNG, BMP, ...) are refused with RTK_ERR_UNSUPPORTED.
#include <cstdio>
#include <cstring>

#include "rtk_internal.hpp"

namespace rtk {
namespace {

const uint8_t kDezigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Fail { int code; std::string msg; };

struct HuffTable {
    // canonical code: for each length the first code and the index of its first symbol
    int32_t first_code[18];
    int32_t first_sym[18];
    int32_t count[18];
    uint8_t symbols[256];
    bool present = false;
    void build(const uint8_t counts[16], const uint8_t *syms, int n) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            first_code[len] = code; first_sym[len] = k; count[len] = counts[len - 1];
            code += counts[len - 1]; k += counts[len - 1];
            if (code > (1 << len)) throw Fail{RTK_ERR_INVALID, "JPEG: bad Huffman code lengths"};
            code <<= 1;
        }
        std::memcpy(symbols, syms, static_cast<size_t>(n));
        present = true;
    }
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool at_marker = false;     // an entropy segment ends at a marker; stb feeds zero bits from there on
    uint8_t marker = 0;
    int bit() {
        if (nbits == 0) {
            uint32_t b = 0;
            if (!at_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    uint8_t c = p < end ? *p : 0;
                    while (c == 0xFF && p < end) { ++p; c = p < end ? *p : 0; }
                    if (p < end) ++p;
                    if (c != 0) { at_marker = true; marker = c; b = 0; }
                }
            }
            acc = b; nbits = 8;
        }
        --nbits;
        return static_cast<int>((acc >> nbits) & 1u);
    }
    int receive(int n) { int v = 0; while (n-- > 0) v = (v << 1) | bit(); return v; }
    int extend_receive(int n) {                       // stbi__extend_receive
        if (n == 0) return 0;
        const int v = receive(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode(const HuffTable &h) {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | bit();
            const int off = code - h.first_code[len];
            if (off >= 0 && off < h.count[len]) return h.symbols[h.first_sym[len] + off];
        }
        throw Fail{RTK_ERR_INVALID, "JPEG: bad Huffman code"};
    }
    void restart() {                                  // stbi__jpeg_reset at the end of a restart interval
        if (!at_marker && p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) p += 2;
        acc = 0; nbits = 0; at_marker = false; marker = 0;
    }
};

inline int f2f(float x) { return static_cast<int>(x * 4096 + 0.5); }       // stbi__f2f: float * int, + double 0.5, truncation
inline int float2fixed(float x) { return static_cast<int>(x * 4096.0f + 0.5f) << 8; }
inline uint8_t clamp_byte(int x) { return static_cast<uint8_t>(x < 0 ? 0 : x > 255 ? 255 : x); }

struct Idct1D { int x0, x1, x2, x3, t0, t1, t2, t3; };

inline Idct1D idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
    int p2 = s2, p3 = s6;
    int p1 = (p2 + p3) * f2f(0.5411961f);
    int t2 = p1 + p3 * f2f(-1.847759065f);
    int t3 = p1 + p2 * f2f(0.765366865f);
    p2 = s0; p3 = s4;
    int t0 = (p2 + p3) * 4096;
    int t1 = (p2 - p3) * 4096;
    Idct1D r;
    r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;
    p3 = t0 + t2; int p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
    const int p5 = (p3 + p4) * f2f(1.175875602f);
    t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f); t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);
    p1 = p5 + p1 * f2f(-0.899976223f);
    p2 = p5 + p2 * f2f(-2.562915447f);
    p3 = p3 * f2f(-1.961570560f);
    p4 = p4 * f2f(-0.390180644f);
    r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
    return r;
}

void idct_block(uint8_t *out, int stride, const short d[64]) {
    int val[64];
    for (int i = 0; i < 8; ++i) {
        if (d[i + 8] == 0 && d[i + 16] == 0 && d[i + 24] == 0 && d[i + 32] == 0 && d[i + 40] == 0 && d[i + 48] == 0 && d[i + 56] == 0) {
            const int dc = d[i] * 4;
            for (int r = 0; r < 8; ++r) val[i + 8 * r] = dc;
        } else {
            Idct1D k = idct_1d(d[i], d[i + 8], d[i + 16], d[i + 24], d[i + 32], d[i + 40], d[i + 48], d[i + 56]);
            k.x0 += 512; k.x1 += 512; k.x2 += 512; k.x3 += 512;
            val[i] = (k.x0 + k.t3) >> 10;      val[i + 56] = (k.x0 - k.t3) >> 10;
            val[i + 8] = (k.x1 + k.t2) >> 10;  val[i + 48] = (k.x1 - k.t2) >> 10;
            val[i + 16] = (k.x2 + k.t1) >> 10; val[i + 40] = (k.x2 - k.t1) >> 10;
            val[i + 24] = (k.x3 + k.t0) >> 10; val[i + 32] = (k.x3 - k.t0) >> 10;
        }
    }
    for (int r = 0; r < 8; ++r) {
        const int *v = val + 8 * r;
        uint8_t *o = out + r * stride;
        Idct1D k = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const int bias = 65536 + (128 << 17);
        k.x0 += bias; k.x1 += bias; k.x2 += bias; k.x3 += bias;
        o[0] = clamp_byte((k.x0 + k.t3) >> 17); o[7] = clamp_byte((k.x0 - k.t3) >> 17);
        o[1] = clamp_byte((k.x1 + k.t2) >> 17); o[6] = clamp_byte((k.x1 - k.t2) >> 17);
        o[2] = clamp_byte((k.x2 + k.t1) >> 17); o[5] = clamp_byte((k.x2 - k.t1) >> 17);
        o[3] = clamp_byte((k.x3 + k.t0) >> 17); o[4] = clamp_byte((k.x3 - k.t0) >> 17);
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int w2 = 0, h2 = 0;          // plane size padded to whole MCUs
    int x = 0, y = 0;            // the component's own size
    int dc_pred = 0;
    std::vector<uint8_t> plane;
};

uint16_t be16(const uint8_t *p) { return static_cast<uint16_t>((p[0] << 8) | p[1]); }

// one output row of a component, expanded by (hs, vs) — what stbi__resample_row_* return for output row j
void resample_row(const Component &c, int hs, int vs, int width, int j, std::vector<uint8_t> &out) {
    out.resize(static_cast<size_t>(width) + 2 * static_cast<size_t>(hs));
    auto row = [&](int k) { if (k < 0) k = 0; if (k > c.y - 1) k = c.y - 1; return c.plane.data() + static_cast<size_t>(k) * c.w2; };
    const int w_lo = (width + hs - 1) / hs;
    if (hs == 1 && vs == 1) { std::memcpy(out.data(), row(j), static_cast<size_t>(width)); return; }
    const uint8_t *nearp = row(vs == 2 ? j >> 1 : j / vs);
    const uint8_t *farp = vs == 2 ? row((j >> 1) + ((j & 1) ? 1 : -1)) : nearp;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < width; ++i) out[i] = static_cast<uint8_t>((3 * nearp[i] + farp[i] + 2) >> 2);
    } else if (hs == 2 && vs == 1) {
        const uint8_t *in = nearp;
        if (w_lo == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0];
        out[1] = static_cast<uint8_t>((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w_lo - 1; ++i) {
            const int n = 3 * in[i] + 2;
            out[i * 2 + 0] = static_cast<uint8_t>((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = static_cast<uint8_t>((n + in[i + 1]) >> 2);
        }
        out[i * 2 + 0] = static_cast<uint8_t>((in[w_lo - 2] * 3 + in[w_lo - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w_lo - 1];
    } else if (hs == 2 && vs == 2) {
        if (w_lo == 1) { out[0] = out[1] = static_cast<uint8_t>((3 * nearp[0] + farp[0] + 2) >> 2); return; }
        int t1 = 3 * nearp[0] + farp[0];
        out[0] = static_cast<uint8_t>((t1 + 2) >> 2);
        for (int i = 1; i < w_lo; ++i) {
            const int t0 = t1;
            t1 = 3 * nearp[i] + farp[i];
            out[i * 2 - 1] = static_cast<uint8_t>((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = static_cast<uint8_t>((3 * t1 + t0 + 8) >> 4);
        }
        out[w_lo * 2 - 1] = static_cast<uint8_t>((t1 + 2) >> 2);
    } else {
        for (int i = 0; i < width; ++i) out[i] = nearp[i / hs];
    }
}

void decode(const uint8_t *data, size_t size, int &width, int &height, int &channels, std::vector<uint8_t> &pixels) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8)
        throw Fail{RTK_ERR_UNSUPPORTED, "bitmap texture is not a JPEG file (only baseline JPEG is decoded)"};
    const uint8_t *p = data + 2, *end = data + size;
    uint16_t dequant[4][64] = {};
    bool have_q[4] = {};
    HuffTable hdc[4], hac[4];
    std::vector<Component> comps;
    int restart_interval = 0, adobe_transform = -1;
    bool jfif = false;
    std::vector<int> order;
    for (;;) {
        while (p < end && *p != 0xFF) ++p;
        while (p < end && *p == 0xFF) ++p;
        if (p >= end) throw Fail{RTK_ERR_INVALID, "JPEG: no scan"};
        const uint8_t m = *p++;
        if (m == 0xD9) throw Fail{RTK_ERR_INVALID, "JPEG: no scan"};
        if (m == 0xC2) throw Fail{RTK_ERR_UNSUPPORTED, "progressive JPEG textures are not decoded (baseline only)"};
        if (m == 0xC1 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC))
            throw Fail{RTK_ERR_UNSUPPORTED, "JPEG: only baseline sequential Huffman coding is decoded"};
        if (p + 2 > end) throw Fail{RTK_ERR_INVALID, "JPEG: truncated"};
        const int L = be16(p);
        if (L < 2 || p + L > end) throw Fail{RTK_ERR_INVALID, "JPEG: bad segment length"};
        const uint8_t *seg = p + 2, *send = p + L;
        if (m == 0xDB) {
            while (seg < send) {
                const int prec = *seg >> 4, t = *seg & 15;
                ++seg;
                if (t > 3 || seg + (prec ? 128 : 64) > send) throw Fail{RTK_ERR_INVALID, "JPEG: bad DQT"};
                for (int i = 0; i < 64; ++i) {
                    dequant[t][kDezigzag[i]] = prec ? be16(seg) : *seg;
                    seg += prec ? 2 : 1;
                }
                have_q[t] = true;
            }
        } else if (m == 0xC4) {
            while (seg < send) {
                const int tc = *seg >> 4, th = *seg & 15;
                if (tc > 1 || th > 3 || seg + 17 > send) throw Fail{RTK_ERR_INVALID, "JPEG: bad DHT"};
                int n = 0;
                for (int i = 0; i < 16; ++i) n += seg[1 + i];
                if (n > 256 || seg + 17 + n > send) throw Fail{RTK_ERR_INVALID, "JPEG: bad DHT"};
                (tc ? hac : hdc)[th].build(seg + 1, seg + 17, n);
                seg += 17 + n;
            }
        } else if (m == 0xC0) {
            if (send - seg < 6) throw Fail{RTK_ERR_INVALID, "JPEG: bad SOF"};
            const int prec = seg[0], n = seg[5];
            height = be16(seg + 1); width = be16(seg + 3);
            if (prec != 8) throw Fail{RTK_ERR_UNSUPPORTED, "JPEG: only 8-bit samples"};
            if (n != 1 && n != 3) throw Fail{RTK_ERR_UNSUPPORTED, "JPEG: only 1- or 3-component files"};
            if (width <= 0 || height <= 0 || send - seg < 6 + 3 * n) throw Fail{RTK_ERR_INVALID, "JPEG: bad SOF"};
            comps.assign(static_cast<size_t>(n), Component{});
            for (int k = 0; k < n; ++k) {
                Component &c = comps[static_cast<size_t>(k)];
                c.id = seg[6 + 3 * k]; c.h = seg[7 + 3 * k] >> 4; c.v = seg[7 + 3 * k] & 15; c.tq = seg[8 + 3 * k];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw Fail{RTK_ERR_INVALID, "JPEG: bad component"};
            }
        } else if (m == 0xDD) {
            if (L != 4) throw Fail{RTK_ERR_INVALID, "JPEG: bad DRI"};
            restart_interval = be16(seg);
        } else if (m == 0xE0 && L >= 7 && std::memcmp(seg, "JFIF\0", 5) == 0) {
            jfif = true;
        } else if (m == 0xEE && L >= 14 && std::memcmp(seg, "Adobe\0", 6) == 0) {
            adobe_transform = seg[11];
        } else if (m == 0xDA) {
            if (comps.empty()) throw Fail{RTK_ERR_INVALID, "JPEG: scan before frame header"};
            const int ns = seg[0];
            if (ns < 1 || ns > static_cast<int>(comps.size()) || send - seg < 1 + 2 * ns + 3) throw Fail{RTK_ERR_INVALID, "JPEG: bad SOS"};
            for (int k = 0; k < ns; ++k) {
                int idx = -1;
                for (size_t i = 0; i < comps.size(); ++i) if (comps[i].id == seg[1 + 2 * k]) idx = static_cast<int>(i);
                if (idx < 0) throw Fail{RTK_ERR_INVALID, "JPEG: bad SOS component"};
                comps[static_cast<size_t>(idx)].hd = seg[2 + 2 * k] >> 4;
                comps[static_cast<size_t>(idx)].ha = seg[2 + 2 * k] & 15;
                order.push_back(idx);
            }
            if (ns != static_cast<int>(comps.size()) && comps.size() != 1)
                throw Fail{RTK_ERR_UNSUPPORTED, "JPEG: multi-scan baseline files are not decoded"};
            p += L;
            break;
        }
        p += L;
    }
    int hmax = 1, vmax = 1;
    for (const Component &c : comps) { if (c.h > hmax) hmax = c.h; if (c.v > vmax) vmax = c.v; }
    const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int mcux = (width + mcu_w - 1) / mcu_w, mcuy = (height + mcu_h - 1) / mcu_h;
    for (Component &c : comps) {
        if (hmax % c.h || vmax % c.v) throw Fail{RTK_ERR_UNSUPPORTED, "JPEG: fractional sampling factors"};
        if (c.hd > 3 || c.ha > 3 || !hdc[c.hd].present || !hac[c.ha].present || !have_q[c.tq]) throw Fail{RTK_ERR_INVALID, "JPEG: missing table"};
        c.x = (width * c.h + hmax - 1) / hmax; c.y = (height * c.v + vmax - 1) / vmax;
        c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;
        c.plane.assign(static_cast<size_t>(c.w2) * static_cast<size_t>(c.h2), 0);
    }
    BitReader br{p, end};
    auto block = [&](Component &c, int bx, int by) {
        short coef[64] = {};
        const uint16_t *dq = dequant[c.tq];
        const int t = br.decode(hdc[c.hd]);
        if (t > 15) throw Fail{RTK_ERR_INVALID, "JPEG: bad DC size"};
        c.dc_pred += t ? br.extend_receive(t) : 0;
        coef[0] = static_cast<short>(c.dc_pred * dq[0]);
        int k = 1;
        do {
            const int rs = br.decode(hac[c.ha]);
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) throw Fail{RTK_ERR_INVALID, "JPEG: coefficient index out of range"};
                const int zig = kDezigzag[k++];
                coef[zig] = static_cast<short>(br.extend_receive(s) * dq[zig]);
            }
        } while (k < 64);
        idct_block(c.plane.data() + static_cast<size_t>(by) * c.w2 + bx, c.w2, coef);
    };
    int todo = restart_interval ? restart_interval : 0x7fffffff;
    auto interval_end = [&]() {
        if (--todo <= 0) {
            br.restart();
            for (Component &c : comps) c.dc_pred = 0;
            todo = restart_interval ? restart_interval : 0x7fffffff;
        }
    };
    if (order.size() == 1) {
        Component &c = comps[static_cast<size_t>(order[0])];
        const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
        for (int j = 0; j < bh; ++j) for (int i = 0; i < bw; ++i) { block(c, i * 8, j * 8); interval_end(); }
    } else {
        for (int j = 0; j < mcuy; ++j) for (int i = 0; i < mcux; ++i) {
            for (int idx : order) {
                Component &c = comps[static_cast<size_t>(idx)];
                for (int y = 0; y < c.v; ++y) for (int x = 0; x < c.h; ++x) block(c, (i * c.h + x) * 8, (j * c.v + y) * 8);
            }
            interval_end();
        }
    }
    channels = static_cast<int>(comps.size());
    pixels.assign(static_cast<size_t>(width) * static_cast<size_t>(height) * static_cast<size_t>(channels), 0);
    if (channels == 1) {
        for (int j = 0; j < height; ++j)
            std::memcpy(pixels.data() + static_cast<size_t>(j) * width, comps[0].plane.data() + static_cast<size_t>(j) * comps[0].w2, static_cast<size_t>(width));
        return;
    }
    // stb: 3 components are RGB when the ids spell "RGB" or an Adobe marker says transform 0 (and there is no JFIF marker), else YCbCr
    const bool is_rgb = (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') || (adobe_transform == 0 && !jfif);
    std::vector<uint8_t> rows[3];
    const int fr = float2fixed(1.40200f), fg_cr = float2fixed(0.71414f), fg_cb = float2fixed(0.34414f), fb = float2fixed(1.77200f);
    for (int j = 0; j < height; ++j) {
        for (int k = 0; k < 3; ++k) resample_row(comps[static_cast<size_t>(k)], hmax / comps[static_cast<size_t>(k)].h, vmax / comps[static_cast<size_t>(k)].v, width, j, rows[k]);
        uint8_t *o = pixels.data() + static_cast<size_t>(j) * width * 3;
        for (int i = 0; i < width; ++i, o += 3) {
            if (is_rgb) { o[0] = rows[0][i]; o[1] = rows[1][i]; o[2] = rows[2][i]; continue; }
            const int y_fixed = (rows[0][i] << 20) + (1 << 19);
            const int cr = rows[2][i] - 128, cb = rows[1][i] - 128;
            int r = y_fixed + cr * fr;
            int g = static_cast<int>(static_cast<uint32_t>(y_fixed + cr * -fg_cr) + (static_cast<uint32_t>(cb * -fg_cb) & 0xffff0000u));
            int b = y_fixed + cb * fb;
            r >>= 20; g >>= 20; b >>= 20;
            o[0] = clamp_byte(r); o[1] = clamp_byte(g); o[2] = clamp_byte(b);
        }
    }
}

}  // namespace

int decode_jpeg(const uint8_t *data, size_t size, int &width, int &height, int &channels, std::vector<uint8_t> &pixels, std::string &err) {
    try {
        decode(data, size, width, height, channels, pixels);
    } catch (const Fail &f) {
        err = f.msg;
        return f.code;
    } catch (const std::bad_alloc &) {
        err = "out of memory decoding a bitmap texture";
        return RTK_ERR_INVALID;
    }
    return RTK_OK;
}

int load_bitmap_file(const std::string &path, int &width, int &height, int &channels, std::vector<uint8_t> &pixels, std::string &err) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open bitmap texture " + path; return RTK_ERR_IO; }
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    std::fclose(f);
    return decode_jpeg(bytes.data(), bytes.size(), width, height, channels, pixels, err);
}

}  // namespace rtk
