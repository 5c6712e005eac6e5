// jpeg_decode.cpp — JPEG reader for ImageTexture ingestion.
//
// The reference decodes textures with stb_image's stbi_load(path, &w, &h, &channels, 0)
// (Source/Texture.cpp:10-21); stb_image is a git submodule that is absent from /root/reference
// (nothings/stb, version unpinned), so this file restates its *published* JPEG pipeline — the choices
// that decide the texel values — rather than any source:
//   * baseline (SOF0/SOF1, 8-bit) and progressive (SOF2) Huffman JPEG, restart intervals, 1 or 3
//     components, any sampling factors; arithmetic coding, 12-bit, lossless and 4-component files fail;
//   * integer "islow"-style IDCT with 12-bit constants, column pass >>10, row pass >>17 (+128 level shift);
//   * chroma upsampling with the 3:1 triangle filters (h2, v2, h2v2), nearest replication otherwise;
//   * YCbCr -> RGB in 20-bit fixed point (1.402, 0.71414, 0.34414, 1.772);
//   * output exactly like req_comp = 0: 1 channel for greyscale files, 3 (RGB) for colour files.
// PARITY UNPINNED against stb_image itself (not available here); tests/test_host_io_cpu.py checks the
// output against Pillow (libjpeg-turbo) within the few grey levels two conforming decoders may differ by.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace Pooraytracer {
namespace {

const uint8_t kDezigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33,
                                    40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                    29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                    47, 55, 62, 63,
                                    // padding so that a corrupt run length cannot index out of the table
                                    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {
    bool present = false;
    // canonical decoding by length (ITU T.81 Annex F.2.2.3): mincode/maxcode/valptr per length
    int32_t maxcode[18];
    int32_t mincode[17];
    int32_t valptr[17];
    uint8_t values[256];
    bool build(const uint8_t counts[16], const uint8_t* vals, int nvals) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += counts[l - 1];
            k += counts[l - 1];
            maxcode[l] = counts[l - 1] ? code - 1 : -1;
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        if (k != nvals || k > 256) return false;
        std::memcpy(values, vals, (size_t)nvals);
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int hd = 0, ha = 0;  // Huffman table selectors of the current scan
    int dc_pred = 0;
    int x = 0, y = 0;    // size in samples: ceil(img * h / hmax)
    int w2 = 0, h2 = 0;  // size padded to whole MCUs
    int bw = 0, bh = 0;  // blocks per row / column of the padded plane
    std::vector<uint8_t> plane;   // w2 * h2 samples
    std::vector<int16_t> coeff;   // progressive: bw * bh * 64
};

struct Decoder {
    const uint8_t* p = nullptr;
    size_t n = 0, pos = 0;
    int img_x = 0, img_y = 0, img_n = 0;
    bool progressive = false, jfif = false;
    int app14_transform = -1;
    int hmax = 1, vmax = 1, mcu_w = 8, mcu_h = 8, mcux = 0, mcuy = 0;
    uint16_t dequant[4][64];
    bool have_q[4] = {false, false, false, false};
    HuffTable hdc[4], hac[4];
    Component comp[4];
    int restart_interval = 0, todo = 0;
    // scan state
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
    // entropy-coded segment reader
    uint32_t code_buffer = 0;
    int code_bits = 0;
    int marker = -1; // marker met inside the entropy data (0xD0.. etc.), -1 if none
    bool nomore = false;
    bool ok = true;

    int u8() { return pos < n ? p[pos++] : (ok = false, 0); }
    int u16() { int a = u8(); return (a << 8) | u8(); }

    void grow() {
        do {
            unsigned b = 0;
            if (!nomore) {
                if (pos >= n) nomore = true;
                else {
                    b = p[pos++];
                    if (b == 0xff) {
                        int c = pos < n ? p[pos++] : 0xd9;
                        while (c == 0xff) c = pos < n ? p[pos++] : 0xd9; // fill bytes
                        if (c != 0) { // a marker ends the entropy-coded data: feed zeros from here on
                            marker = c;
                            nomore = true;
                            b = 0;
                        }
                    }
                }
            }
            code_buffer |= b << (24 - code_bits);
            code_bits += 8;
        } while (code_bits <= 24);
    }
    int get_bit() {
        if (code_bits < 1) grow();
        int b = (int)(code_buffer >> 31);
        code_buffer <<= 1;
        code_bits--;
        return b;
    }
    int get_bits(int k) {
        if (k == 0) return 0;
        if (code_bits < k) grow();
        int v = (int)(code_buffer >> (32 - k));
        code_buffer <<= k;
        code_bits -= k;
        return v;
    }
    int extend_receive(int k) { // T.81 F.2.2.1 EXTEND
        if (k == 0) return 0;
        int v = get_bits(k);
        return v < (1 << (k - 1)) ? v - (1 << k) + 1 : v;
    }
    int decode(const HuffTable& h) {
        if (!h.present) { ok = false; return 0; }
        if (code_bits < 16) grow();
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (int)(code_buffer >> (32 - l));
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
                code_buffer <<= l;
                code_bits -= l;
                return h.values[h.valptr[l] + code - h.mincode[l]];
            }
        }
        ok = false;
        return 0;
    }
    void reset_entropy() {
        code_bits = 0;
        code_buffer = 0;
        nomore = false;
        marker = -1;
        for (auto& c : comp) c.dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    // ---- block decoders
    bool block_baseline(int16_t data[64], Component& c) {
        std::memset(data, 0, 64 * sizeof(int16_t));
        const uint16_t* dq = dequant[c.tq];
        int t = decode(hdc[c.hd]);
        if (!ok || t > 15) return false;
        int diff = t ? extend_receive(t) : 0;
        c.dc_pred += diff;
        data[0] = (int16_t)(c.dc_pred * dq[0]);
        int k = 1;
        do {
            int rs = decode(hac[c.ha]);
            if (!ok) return false;
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                int zig = kDezigzag[k++];
                data[zig] = (int16_t)(extend_receive(s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }
    bool block_prog_dc(int16_t data[64], Component& c) {
        if (spec_end != 0) return false;
        if (succ_high == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            int t = decode(hdc[c.hd]);
            if (!ok || t > 15) return false;
            int diff = t ? extend_receive(t) : 0;
            c.dc_pred += diff;
            data[0] = (int16_t)(c.dc_pred * (1 << succ_low));
        } else if (get_bit()) {
            data[0] = (int16_t)(data[0] + (1 << succ_low));
        }
        return true;
    }
    bool block_prog_ac(int16_t data[64], Component& c) {
        if (spec_start == 0) return false;
        const HuffTable& h = hac[c.ha];
        if (succ_high == 0) {
            if (eob_run) { --eob_run; return true; }
            int k = spec_start;
            do {
                int rs = decode(h);
                if (!ok) return false;
                int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    int zig = kDezigzag[k++];
                    data[zig] = (int16_t)(extend_receive(s) * (1 << succ_low));
                }
            } while (k <= spec_end);
        } else { // refinement pass (T.81 G.1.2.3)
            const int16_t bit = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t* q) {
                if (get_bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) {
                    int16_t* q = &data[kDezigzag[k]];
                    if (*q != 0) refine(q);
                }
            } else {
                int k = spec_start;
                do {
                    int rs = decode(h);
                    if (!ok) return false;
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += get_bits(r);
                            r = 64; // run to the end of the band
                        }
                        // r == 15: skip 16 zero coefficients (refining the non-zero ones passed on the way)
                    } else {
                        if (s != 1) return false;
                        s = get_bit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        int16_t* q = &data[kDezigzag[k++]];
                        if (*q != 0) refine(q);
                        else {
                            if (r == 0) { *q = (int16_t)s; break; }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
        return true;
    }

    // ---- integer IDCT, 8x8, output stride `stride`
    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }
    static void idct(uint8_t* out, int stride, const int16_t d[64]) {
        auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
        static const int c0541 = f2f(0.5411961), c1847 = -f2f(1.847759065), c0765 = f2f(0.765366865),
                         c1175 = f2f(1.175875602), c0298 = f2f(0.298631336), c2053 = f2f(2.053119869),
                         c3072 = f2f(3.072711026), c1501 = f2f(1.501321110), c0899 = -f2f(0.899976223),
                         c2562 = -f2f(2.562915447), c1961 = -f2f(1.961570560), c0390 = -f2f(0.390180644);
        int val[64];
        int x0, x1, x2, x3, t0, t1, t2, t3;
        auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
            int p1, p2, p3, p4, p5;
            p2 = s2; p3 = s6;
            p1 = (p2 + p3) * c0541;
            t2 = p1 + p3 * c1847;
            t3 = p1 + p2 * c0765;
            p2 = s0; p3 = s4;
            t0 = (p2 + p3) * 4096;
            t1 = (p2 - p3) * 4096;
            x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
            t0 = s7; t1 = s5; t2 = s3; t3 = s1;
            p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
            p5 = (p3 + p4) * c1175;
            t0 = t0 * c0298; t1 = t1 * c2053; t2 = t2 * c3072; t3 = t3 * c1501;
            p1 = p5 + p1 * c0899;
            p2 = p5 + p2 * c2562;
            p3 = p3 * c1961;
            p4 = p4 * c0390;
            t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        };
        for (int i = 0; i < 8; ++i) { // columns
            const int16_t* c = d + i;
            int* v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                int dc = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
            } else {
                pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
                v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
                v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
                v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) { // rows
            const int* v = val + i * 8;
            uint8_t* o = out + i * stride;
            pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const int bias = 65536 + (128 << 17);
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
    }

    // ---- segments
    bool read_dqt(int len) {
        while (len > 0) {
            int q = u8();
            int prec = q >> 4, t = q & 15;
            if (prec > 1 || t > 3) return false;
            for (int i = 0; i < 64; ++i) dequant[t][kDezigzag[i]] = (uint16_t)(prec ? u16() : u8());
            have_q[t] = true;
            len -= prec ? 129 : 65;
        }
        return len == 0 && ok;
    }
    bool read_dht(int len) {
        while (len > 0) {
            int q = u8();
            int tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) return false;
            uint8_t counts[16], vals[256];
            int total = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)u8(); total += counts[i]; }
            if (total > 256) return false;
            for (int i = 0; i < total; ++i) vals[i] = (uint8_t)u8();
            if (!ok || !(tc == 0 ? hdc[th] : hac[th]).build(counts, vals, total)) return false;
            len -= 17 + total;
        }
        return len == 0;
    }
    bool read_sof(int len) {
        if (img_n) return false; // second frame header
        int prec = u8();
        img_y = u16();
        img_x = u16();
        img_n = u8();
        if (prec != 8 || img_y <= 0 || img_x <= 0) return false;
        if (img_n != 1 && img_n != 3) return false;
        if (len != 6 + 3 * img_n) return false;
        if ((uint64_t)img_x * (uint64_t)img_y > (1ull << 28)) return false;
        for (int i = 0; i < img_n; ++i) {
            comp[i].id = u8();
            int q = u8();
            comp[i].h = q >> 4;
            comp[i].v = q & 15;
            comp[i].tq = u8();
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return false;
            if (comp[i].h > hmax) hmax = comp[i].h;
            if (comp[i].v > vmax) vmax = comp[i].v;
        }
        for (int i = 0; i < img_n; ++i)
            if (hmax % comp[i].h || vmax % comp[i].v) return false;
        mcu_w = hmax * 8;
        mcu_h = vmax * 8;
        mcux = (img_x + mcu_w - 1) / mcu_w;
        mcuy = (img_y + mcu_h - 1) / mcu_h;
        for (int i = 0; i < img_n; ++i) {
            Component& c = comp[i];
            c.x = (img_x * c.h + hmax - 1) / hmax;
            c.y = (img_y * c.v + vmax - 1) / vmax;
            c.w2 = mcux * c.h * 8;
            c.h2 = mcuy * c.v * 8;
            c.bw = c.w2 / 8;
            c.bh = c.h2 / 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coeff.assign((size_t)c.bw * c.bh * 64, 0);
        }
        return ok;
    }
    bool read_sos(int len) {
        scan_n = u8();
        if (scan_n < 1 || scan_n > img_n || len != 4 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; ++i) {
            int id = u8(), q = u8(), which = -1;
            for (int k = 0; k < img_n; ++k)
                if (comp[k].id == id) which = k;
            if (which < 0) return false;
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return false;
            order[i] = which;
        }
        spec_start = u8();
        spec_end = u8();
        int a = u8();
        succ_high = a >> 4;
        succ_low = a & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return false;
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) return false;
            spec_end = 63;
        }
        return ok;
    }

    bool restart_check() { // after each MCU / block: handle DRI
        if (--todo <= 0) {
            if (code_bits < 24) grow();
            if (marker < 0xd0 || marker > 0xd7) return false; // no RSTn: the scan ends here
            reset_entropy();
        }
        return true;
    }

    bool decode_scan() {
        reset_entropy();
        for (int i = 0; i < scan_n; ++i)
            if (!have_q[comp[order[i]].tq]) return false;
        int16_t data[64];
        if (!progressive) {
            if (scan_n == 1) {
                Component& c = comp[order[0]];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; ++j)
                    for (int i = 0; i < w; ++i) {
                        if (!block_baseline(data, c)) return false;
                        idct(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                        if (!restart_check()) return true;
                    }
            } else {
                for (int j = 0; j < mcuy; ++j)
                    for (int i = 0; i < mcux; ++i) {
                        for (int k = 0; k < scan_n; ++k) {
                            Component& c = comp[order[k]];
                            for (int y = 0; y < c.v; ++y)
                                for (int x = 0; x < c.h; ++x) {
                                    const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                                    if (!block_baseline(data, c)) return false;
                                    idct(c.plane.data() + (size_t)c.w2 * y2 + x2, c.w2, data);
                                }
                        }
                        if (!restart_check()) return true;
                    }
            }
        } else {
            if (scan_n == 1) {
                Component& c = comp[order[0]];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; ++j)
                    for (int i = 0; i < w; ++i) {
                        int16_t* blk = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.bw);
                        if (spec_start == 0 ? !block_prog_dc(blk, c) : !block_prog_ac(blk, c)) return false;
                        if (!restart_check()) return true;
                    }
            } else {
                if (spec_start != 0) return false; // AC scans are never interleaved
                for (int j = 0; j < mcuy; ++j)
                    for (int i = 0; i < mcux; ++i) {
                        for (int k = 0; k < scan_n; ++k) {
                            Component& c = comp[order[k]];
                            for (int y = 0; y < c.v; ++y)
                                for (int x = 0; x < c.h; ++x) {
                                    const size_t x2 = (size_t)i * c.h + x, y2 = (size_t)j * c.v + y;
                                    if (!block_prog_dc(c.coeff.data() + 64 * (x2 + y2 * c.bw), c)) return false;
                                }
                        }
                        if (!restart_check()) return true;
                    }
            }
        }
        return true;
    }

    void finish_progressive() {
        for (int k = 0; k < img_n; ++k) {
            Component& c = comp[k];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    int16_t* blk = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.bw);
                    for (int q = 0; q < 64; ++q) blk[q] = (int16_t)(blk[q] * dequant[c.tq][q]);
                    idct(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, blk);
                }
        }
    }

    // next marker after the entropy-coded data (or at the current position between segments)
    int next_marker() {
        if (marker >= 0) { int m = marker; marker = -1; return m; }
        while (pos + 1 < n) {
            if (p[pos] != 0xff) { ++pos; continue; }
            int c = p[pos + 1];
            if (c == 0xff) { ++pos; continue; }
            if (c == 0) { pos += 2; continue; }
            pos += 2;
            return c;
        }
        return -1;
    }

    bool parse() {
        if (n < 4 || p[0] != 0xff || p[1] != 0xd8) return false;
        pos = 2;
        bool seen_scan = false;
        for (;;) {
            int m = next_marker();
            if (m < 0) return seen_scan; // truncated file: keep what was decoded, like a missing EOI
            if (m == 0xd9) break;
            if (m >= 0xd0 && m <= 0xd7) continue; // stray RSTn
            if (m == 0x01) continue;              // TEM has no length
            int len = u16() - 2;
            if (!ok || len < 0 || pos + (size_t)len > n) return seen_scan;
            const size_t seg_end = pos + (size_t)len;
            switch (m) {
            case 0xc0: case 0xc1: case 0xc2:
                progressive = m == 0xc2;
                if (!read_sof(len)) return false;
                break;
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                return false; // lossless / hierarchical / arithmetic coding
            case 0xc4: if (!read_dht(len)) return false; break;
            case 0xdb: if (!read_dqt(len)) return false; break;
            case 0xdd:
                if (len != 2) return false;
                restart_interval = u16();
                break;
            case 0xe0:
                if (len >= 5 && !std::memcmp(p + pos, "JFIF\0", 5)) jfif = true;
                break;
            case 0xee:
                if (len >= 12 && !std::memcmp(p + pos, "Adobe\0", 6)) app14_transform = p[pos + 11];
                break;
            case 0xda:
                if (!img_n || !read_sos(len)) return false;
                pos = seg_end;
                if (!decode_scan()) return false;
                seen_scan = true;
                continue; // position is inside / after the entropy data; next_marker() resynchronises
            default: break; // APPn, COM, DNL, ...: skipped
            }
            pos = seg_end;
        }
        return seen_scan;
    }

    // ---- upsampling rows (triangle filters) and colour conversion
    static void row_h2(uint8_t* out, const uint8_t* in, int w) {
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) {
            int t = 3 * in[i] + 2;
            out[i * 2] = (uint8_t)((t + in[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((t + in[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
    }
    static void row_v2(uint8_t* out, const uint8_t* nr, const uint8_t* fr, int w) {
        for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2);
    }
    static void row_hv2(uint8_t* out, const uint8_t* nr, const uint8_t* fr, int w) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2); return; }
        int t1 = 3 * nr[0] + fr[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            int t0 = t1;
            t1 = 3 * nr[i] + fr[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    }

    bool output(int& w, int& h, int& channels, std::vector<unsigned char>& px) {
        if (progressive) finish_progressive();
        w = img_x;
        h = img_y;
        channels = img_n == 3 ? 3 : 1;
        px.assign((size_t)w * h * channels, 0);
        bool is_rgb = false;
        if (img_n == 3) {
            const bool ids_rgb = comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B';
            is_rgb = ids_rgb || (app14_transform == 0 && !jfif);
        }
        struct Up { int hs, vs, ystep, wl, ypos; const uint8_t *l0, *l1; std::vector<uint8_t> line; };
        Up up[3];
        for (int k = 0; k < img_n; ++k) {
            Up& r = up[k];
            r.hs = hmax / comp[k].h;
            r.vs = vmax / comp[k].v;
            r.ystep = r.vs >> 1;
            r.wl = (img_x + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.l0 = r.l1 = comp[k].plane.data();
            r.line.assign((size_t)img_x + 8, 0);
        }
        const int f_r = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, f_g1 = ((int)(0.71414f * 4096.0f + 0.5f)) << 8,
                  f_g2 = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, f_b = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
        for (int j = 0; j < img_y; ++j) {
            const uint8_t* rows[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < img_n; ++k) {
                Up& r = up[k];
                const bool bot = r.ystep >= (r.vs >> 1);
                const uint8_t* nr = bot ? r.l1 : r.l0;
                const uint8_t* fr = bot ? r.l0 : r.l1;
                if (r.hs == 1 && r.vs == 1) rows[k] = nr;
                else {
                    r.line.resize((size_t)r.wl * r.hs + 8);
                    if (r.hs == 1 && r.vs == 2) row_v2(r.line.data(), nr, fr, r.wl);
                    else if (r.hs == 2 && r.vs == 1) row_h2(r.line.data(), nr, r.wl);
                    else if (r.hs == 2 && r.vs == 2) row_hv2(r.line.data(), nr, fr, r.wl);
                    else
                        for (int i = 0; i < r.wl; ++i)
                            for (int q = 0; q < r.hs; ++q) r.line[(size_t)i * r.hs + q] = nr[i];
                    rows[k] = r.line.data();
                }
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.l0 = r.l1;
                    if (++r.ypos < comp[k].y) r.l1 += comp[k].w2;
                }
            }
            unsigned char* out = px.data() + (size_t)j * w * channels;
            if (img_n == 1) std::memcpy(out, rows[0], (size_t)w);
            else if (is_rgb)
                for (int i = 0; i < w; ++i) { out[3 * i] = rows[0][i]; out[3 * i + 1] = rows[1][i]; out[3 * i + 2] = rows[2][i]; }
            else
                for (int i = 0; i < w; ++i) {
                    const int yf = (rows[0][i] << 20) + (1 << 19);
                    const int cb = rows[1][i] - 128, cr = rows[2][i] - 128;
                    int r = yf + cr * f_r;
                    int g = yf + cr * -f_g1 + (int)((unsigned)(cb * -f_g2) & 0xffff0000u);
                    int b = yf + cb * f_b;
                    out[3 * i] = clamp8(r >> 20);
                    out[3 * i + 1] = clamp8(g >> 20);
                    out[3 * i + 2] = clamp8(b >> 20);
                }
        }
        return true;
    }
};

} // namespace

// stbi_load(path, &w, &h, &channels, 0) for JPEG files; false = not a JPEG this reader handles.
bool load_jpeg(const std::string& path, int& w, int& h, int& channels, std::vector<unsigned char>& pixels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Decoder d;
    d.p = buf.data();
    d.n = buf.size();
    std::memset(d.dequant, 0, sizeof(d.dequant));
    if (!d.parse() || !d.img_n) return false;
    return d.output(w, h, channels, pixels);
}

} // namespace Pooraytracer
