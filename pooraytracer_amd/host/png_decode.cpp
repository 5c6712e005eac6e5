// png_decode.cpp — minimal PNG reader for ImageTexture ingestion (the reference
// uses stb_image, Source/Texture.cpp:10-21).  Supports what texture assets use in practice: 8-bit
// greyscale / grey+alpha / RGB / RGBA, non-interlaced; zlib streams with stored, fixed-Huffman and
// dynamic-Huffman blocks.  Returns texels exactly like stbi_load(path, &w, &h, &channels, 0): interleaved,
// row 0 first, file's own channel count.  Anything else (16-bit, palette, interlaced, ...) fails
// and the caller falls back to the reference's failed-load behaviour.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace Pooraytracer {
namespace {

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    bool ok = true;
    int bit() {
        if (cnt == 0) {
            if (pos >= n) { ok = false; return 0; }
            buf = p[pos++];
            cnt = 8;
        }
        int b = buf & 1;
        buf >>= 1;
        cnt--;
        return b;
    }
    uint32_t bits(int k) {
        uint32_t v = 0;
        for (int i = 0; i < k; ++i) v |= (uint32_t)bit() << i;
        return v;
    }
    void align() { cnt = 0; }
};

struct Huffman { // canonical Huffman decoding by code length counts (RFC 1951 §3.2.2)
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const uint8_t* len, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= br.bit();
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
            if (!br.ok) return -1;
        }
        return -1;
    }
};

bool inflate(const std::vector<uint8_t>& z, std::vector<uint8_t>& out) {
    if (z.size() < 6) return false;
    BitReader br{z.data() + 2, z.size() - 2}; // skip the 2-byte zlib header
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int last;
    do {
        last = br.bit();
        const int type = (int)br.bits(2);
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lens[i] = 8;
                for (; i < 256; ++i) lens[i] = 9;
                for (; i < 280; ++i) lens[i] = 7;
                for (; i < 288; ++i) lens[i] = 8;
                lit.build(lens, 288);
                for (i = 0; i < 30; ++i) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman lencode;
                lencode.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return false;
                            val = lens[idx - 1];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                lit.build(lens, nlen);
                dist.build(lens + nlen, ndist);
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0 || !br.ok) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    const int len = lbase[sym] + (int)br.bits(lext[sym]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (d > out.size()) return false;
                    for (int i = 0; i < len; ++i) out.push_back(out[out.size() - d]);
                }
            }
        } else {
            return false;
        }
        if (!br.ok) return false;
    } while (!last);
    return true;
}

} // namespace

bool load_png(const std::string& path, int& w, int& h, int& channels, std::vector<unsigned char>& pixels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
    size_t pos = 8;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> z;
    w = h = 0;
    while (pos + 12 <= d.size()) {
        const uint32_t len = be32(pos);
        const std::string tag(reinterpret_cast<const char*>(&d[pos + 4]), 4);
        if (pos + 12 + len > d.size()) return false;
        const uint8_t* body = &d[pos + 8];
        if (tag == "IHDR") {
            w = (int)be32(pos + 8);
            h = (int)be32(pos + 12);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (tag == "IDAT") {
            z.insert(z.end(), body, body + len);
        } else if (tag == "IEND") {
            break;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) return false;
    channels = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!channels) return false;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * channels + 1));
    if (!inflate(z, raw)) return false;
    const size_t stride = (size_t)w * channels;
    if (raw.size() < (size_t)h * (stride + 1)) return false;
    pixels.assign((size_t)h * stride, 0);
    auto paeth = [](int a, int b, int c) {
        const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
        return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
    };
    for (int y = 0; y < h; ++y) { // PNG filter types 0-4 (RFC 2083 §6)
        const uint8_t* src = &raw[(size_t)y * (stride + 1)];
        uint8_t* dst = &pixels[(size_t)y * stride];
        const uint8_t* up = y ? dst - stride : nullptr;
        const int ft = src[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)channels ? dst[x - channels] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= (size_t)channels) ? up[x - channels] : 0;
            int v = src[x + 1];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            dst[x] = (uint8_t)v;
        }
    }
    return true;
}

} // namespace Pooraytracer
