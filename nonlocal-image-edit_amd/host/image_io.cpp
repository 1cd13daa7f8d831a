// Image file I/O for the enhance CLI (stand-in for OpenCV highgui/imgcodecs, which the reference
// uses at src/enhance.cpp:33,47).  No arithmetic of the filter lives here.  BMP / PPM / PNG here, JPEG in jpeg.cpp.
#include "nle/image_io.hpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace nle {

Image read_jpeg(const std::vector<unsigned char>& b);                                // jpeg.cpp
bool write_jpeg(std::vector<unsigned char>* out, const Image& bgr, int quality);    // jpeg.cpp

namespace {

bool read_file(const std::string& path, std::vector<unsigned char>* out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n <= 0) {
        std::fclose(f);
        return false;
    }
    out->resize((size_t)n);
    const size_t got = std::fread(out->data(), 1, (size_t)n, f);
    std::fclose(f);
    return got == (size_t)n;
}

uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

Image read_bmp(const std::vector<unsigned char>& b) {
    if (b.size() < 54 || b[0] != 'B' || b[1] != 'M') return Image();
    const uint32_t off = rd32(&b[10]), hdr = rd32(&b[14]);
    if (hdr < 40) return Image();
    const int32_t w = (int32_t)rd32(&b[18]), hraw = (int32_t)rd32(&b[22]);
    const uint16_t bpp = rd16(&b[28]);
    const uint32_t comp = rd32(&b[30]);
    // dimensions are bounded before any arithmetic on them: a crafted header must not wrap the size check below or
    // ask for a huge allocation (hraw == INT_MIN cannot be negated)
    constexpr int32_t kMaxDim = 65535;
    if (w <= 0 || w > kMaxDim || hraw == 0 || hraw < -kMaxDim || hraw > kMaxDim || (bpp != 24 && bpp != 32) ||
        (comp != 0 && comp != 3))
        return Image();
    const int h = hraw < 0 ? -hraw : hraw;
    const size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
    if ((size_t)off > b.size() || stride * (size_t)h > b.size() - (size_t)off) return Image();
    Image img(h, w, NLE_8U, 3);
    for (int r = 0; r < h; ++r) {
        const unsigned char* src = &b[off + stride * (size_t)(hraw < 0 ? r : h - 1 - r)];
        unsigned char* dst = img.ptr<unsigned char>(r);
        for (int c = 0; c < w; ++c) {
            dst[3 * c + 0] = src[(bpp / 8) * c + 0];
            dst[3 * c + 1] = src[(bpp / 8) * c + 1];
            dst[3 * c + 2] = src[(bpp / 8) * c + 2];
        }
    }
    return img;
}

Image read_ppm(const std::vector<unsigned char>& b) {
    size_t pos = 2;
    auto next_int = [&]() -> long {
        while (pos < b.size()) {
            if (b[pos] == '#') {
                while (pos < b.size() && b[pos] != '\n') ++pos;
            } else if (b[pos] == ' ' || b[pos] == '\n' || b[pos] == '\r' || b[pos] == '\t') {
                ++pos;
            } else {
                break;
            }
        }
        long v = 0;
        bool any = false;
        while (pos < b.size() && b[pos] >= '0' && b[pos] <= '9') {
            v = v * 10 + (b[pos++] - '0');
            any = true;
        }
        return any ? v : -1;
    };
    const long w = next_int(), h = next_int(), mx = next_int();
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || mx != 255) return Image();
    ++pos;  // single whitespace after maxval
    if (pos + (size_t)w * h * 3 > b.size()) return Image();
    Image img((int)h, (int)w, NLE_8U, 3);
    unsigned char* d = img.ptr<unsigned char>();
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        d[3 * i + 0] = b[pos + 3 * i + 2];
        d[3 * i + 1] = b[pos + 3 * i + 1];
        d[3 * i + 2] = b[pos + 3 * i + 0];
    }
    return img;
}

// ---- PNG reader: zlib inflate (stored, fixed and dynamic Huffman blocks) + the five scanline filters.  8-bit and
// 16-bit samples (high byte kept), grey / RGB / palette / with alpha (dropped), non-interlaced -- what `enhance` writes
// and what common tools produce.  Interlaced files are refused.
struct BitReader {
    const unsigned char* p;
    size_t n, pos = 0;
    uint32_t bits = 0;
    int nbits = 0;
    bool ok = true;
    uint32_t get(int k) {
        while (nbits < k) {
            if (pos >= n) {
                ok = false;
                return 0;
            }
            bits |= (uint32_t)p[pos++] << nbits;
            nbits += 8;
        }
        const uint32_t v = k == 32 ? bits : (bits & ((1u << k) - 1));
        bits = k == 32 ? 0 : bits >> k;
        nbits -= k;
        return v;
    }
};

struct Huff {  // canonical Huffman decoding table (counts per length + symbols in code order)
    uint16_t count[16] = {0}, symbol[288] = {0};
    bool build(const unsigned char* len, int n) {
        for (int i = 0; i < 16; ++i) count[i] = 0;
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)br.get(1);
            if (!br.ok) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate_zlib(const std::vector<unsigned char>& z, std::vector<unsigned char>* out, size_t expected) {
    if (z.size() < 6 || (z[0] & 0x0f) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return false;
    BitReader br{z.data() + 2, z.size() - 2};
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out->clear();
    // deflate expands by at most 1032 : 1 (a 258-byte match from a 2-bit code): a header promising more than the stream
    // can hold is rejected before anything is reserved (a crafted IHDR asked for 34 GB from a 100-byte file)
    if (expected > z.size() * 1032 + 64) return false;
    out->reserve(expected);
    for (;;) {
        const uint32_t last = br.get(1), type = br.get(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.bits = 0;
            br.nbits = 0;  // skip to the byte boundary
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n || out->size() + len > expected) return false;
            out->insert(out->end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huff lit, dist;
            unsigned char lens[320];
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
                const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) return false;
                static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                unsigned char cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (unsigned char)br.get(3);
                Huff clh;
                if (!br.ok || !clh.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lens[idx++] = (unsigned char)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return false;
                            val = lens[idx - 1];
                            rep = 3 + (int)br.get(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)br.get(3);
                        } else {
                            rep = 11 + (int)br.get(7);
                        }
                        if (!br.ok || idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (unsigned char)val;
                    }
                }
                if (lens[256] == 0 || !lit.build(lens, nlen) || !dist.build(lens + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) {
                    if (out->size() >= expected) return false;
                    out->push_back((unsigned char)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    if (sym > 285) return false;
                    const int len = lbase[sym - 257] + (int)br.get(lext[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds > 29) return false;
                    const size_t d = dbase[ds] + br.get(dext[ds]);
                    if (!br.ok || d > out->size() || out->size() + (size_t)len > expected) return false;
                    for (int i = 0; i < len; ++i) out->push_back((*out)[out->size() - d]);
                }
            }
        } else {
            return false;
        }
        if (last) break;
    }
    if (out->size() != expected) return false;
    // the zlib trailer: Adler-32 of the inflated bytes, big endian, right after the last block (byte aligned)
    size_t tail = br.pos - (size_t)(br.nbits / 8);  // whole bytes still in the bit buffer were not consumed
    if (tail + 4 > br.n) return false;
    uint32_t a = 1, b2 = 0;
    for (size_t i = 0; i < out->size();) {
        const size_t stop = std::min(out->size(), i + 5552);  // largest run before the sums can overflow 32 bits
        for (; i < stop; ++i) {
            a += (*out)[i];
            b2 += a;
        }
        a %= 65521u;
        b2 %= 65521u;
    }
    const uint32_t want = ((uint32_t)br.p[tail] << 24) | ((uint32_t)br.p[tail + 1] << 16) | ((uint32_t)br.p[tail + 2] << 8) | br.p[tail + 3];
    return ((b2 << 16) | a) == want;
}

uint32_t rd32be(const unsigned char* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

Image read_png(const std::vector<unsigned char>& b) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (b.size() < 33 || std::memcmp(b.data(), sig, 8) != 0) return Image();
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1;
    std::vector<unsigned char> idat, plte;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= b.size()) {
        const uint32_t len = rd32be(&b[pos]);
        if (len > b.size() || pos + 12 + len > b.size()) return Image();
        const unsigned char* type = &b[pos + 4];
        const unsigned char* data = &b[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return Image();
            w = rd32be(data);
            h = rd32be(data + 4);
            depth = data[8];
            ctype = data[9];
            if (data[10] != 0 || data[11] != 0 || data[12] != 0) return Image();  // interlaced: not supported
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 65535 || h > 65535) return Image();
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: return Image();
    }
    if (ctype == 3 ? !(depth == 1 || depth == 2 || depth == 4 || depth == 8) : !(depth == 8 || depth == 16)) return Image();
    if (ctype == 3 && plte.size() < 3) return Image();
    const size_t bits_pp = (size_t)ch * depth, bpp = (bits_pp + 7) / 8, stride = ((size_t)w * bits_pp + 7) / 8;
    std::vector<unsigned char> raw;
    if (!inflate_zlib(idat, &raw, (stride + 1) * (size_t)h)) return Image();
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    Image img((int)h, (int)w, NLE_8U, 3);
    for (uint32_t r = 0; r < h; ++r) {
        const unsigned char* line = &raw[(stride + 1) * (size_t)r];
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = up; break;
                case 3: pred = (a + up) >> 1; break;
                case 4: {
                    const int pa = std::abs(up - c), pb = std::abs(a - c), pc = std::abs(a + up - 2 * c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? up : c);
                } break;
                default: return Image();
            }
            cur[i] = (unsigned char)(line[1 + i] + pred);
        }
        unsigned char* dst = img.ptr<unsigned char>((int)r);
        const int step = depth == 16 ? 2 : 1;
        for (uint32_t x = 0; x < w; ++x) {
            unsigned char R, G, B;
            if (ctype == 3) {
                const size_t bit = (size_t)x * depth;
                const unsigned idx = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
                if ((size_t)idx * 3 + 2 >= plte.size()) return Image();
                R = plte[idx * 3], G = plte[idx * 3 + 1], B = plte[idx * 3 + 2];
            } else {
                const unsigned char* px = &cur[(size_t)x * ch * step];
                if (ch <= 2) R = G = B = px[0];
                else R = px[0], G = px[step], B = px[2 * step];
            }
            dst[3 * x + 0] = B;  // BGR like cv::imread
            dst[3 * x + 1] = G;
            dst[3 * x + 2] = R;
        }
        prev.swap(cur);
    }
    return img;
}

void wr32(std::vector<unsigned char>& v, uint32_t x) {
    for (int i = 0; i < 4; ++i) v.push_back((unsigned char)(x >> (8 * i)));
}
void wr32be(std::vector<unsigned char>& v, uint32_t x) {
    for (int i = 3; i >= 0; --i) v.push_back((unsigned char)(x >> (8 * i)));
}
void wr16(std::vector<unsigned char>& v, uint16_t x) {
    v.push_back((unsigned char)x);
    v.push_back((unsigned char)(x >> 8));
}

uint32_t crc32_of(const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

void png_chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data) {
    wr32be(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    wr32be(out, crc32_of(&out[start], out.size() - start));
}

bool write_all(const std::string& path, const std::vector<unsigned char>& v) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(v.data(), 1, v.size(), f) == v.size();
    std::fclose(f);
    return ok;
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i) {
        char a = s[s.size() - n + i];
        if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
        if (a != suf[i]) return false;
    }
    return true;
}

}  // namespace

Image imread(const std::string& path) {
    std::vector<unsigned char> b;
    if (!read_file(path, &b) || b.size() < 2) return Image();
    if (b[0] == 'B' && b[1] == 'M') return read_bmp(b);
    if (b[0] == 'P' && b[1] == '6') return read_ppm(b);
    if (b[0] == 0x89 && b[1] == 'P') return read_png(b);
    if (b[0] == 0xff && b[1] == 0xd8) return read_jpeg(b);
    return Image();
}

bool imwrite(const std::string& path, const Image& img) {
    if (img.empty() || img.channels() != 3 || img.depth() != NLE_8U) return false;
    const int w = img.cols, h = img.rows;
    std::vector<unsigned char> out;
    if (ends_with(path, ".ppm")) {
        char hdr[64];
        const int n = std::snprintf(hdr, sizeof hdr, "P6\n%d %d\n255\n", w, h);
        out.insert(out.end(), hdr, hdr + n);
        const unsigned char* s = img.ptr<unsigned char>();
        for (size_t i = 0; i < img.total(); ++i) {
            out.push_back(s[3 * i + 2]);
            out.push_back(s[3 * i + 1]);
            out.push_back(s[3 * i + 0]);
        }
        return write_all(path, out);
    }
    if (ends_with(path, ".jpg") || ends_with(path, ".jpeg")) {  // cv::imwrite's default quality
        if (!write_jpeg(&out, img, 95)) return false;
        return write_all(path, out);
    }
    if (ends_with(path, ".png")) {
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        out.insert(out.end(), sig, sig + 8);
        std::vector<unsigned char> ihdr;
        wr32be(ihdr, (uint32_t)w);
        wr32be(ihdr, (uint32_t)h);
        const unsigned char tail[5] = {8, 2, 0, 0, 0};  // 8-bit RGB, deflate, no filter, no interlace
        ihdr.insert(ihdr.end(), tail, tail + 5);
        png_chunk(out, "IHDR", ihdr);
        // raw scanlines: filter byte 0 + RGB
        std::vector<unsigned char> raw;
        raw.reserve((size_t)h * (1 + 3 * (size_t)w));
        for (int r = 0; r < h; ++r) {
            raw.push_back(0);
            const unsigned char* s = img.ptr<unsigned char>(r);
            for (int c = 0; c < w; ++c) {
                raw.push_back(s[3 * c + 2]);
                raw.push_back(s[3 * c + 1]);
                raw.push_back(s[3 * c + 0]);
            }
        }
        // zlib stream of "stored" deflate blocks (<= 65535 bytes each) + Adler-32
        std::vector<unsigned char> z;
        z.push_back(0x78);
        z.push_back(0x01);
        uint32_t a = 1, b2 = 0;
        size_t pos = 0;
        while (pos < raw.size() || raw.empty()) {
            const size_t n = std::min<size_t>(65535, raw.size() - pos);
            z.push_back(pos + n >= raw.size() ? 1 : 0);
            wr16(z, (uint16_t)n);
            wr16(z, (uint16_t)~n);
            z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
            for (size_t i = 0; i < n; ++i) {
                a = (a + raw[pos + i]) % 65521;
                b2 = (b2 + a) % 65521;
            }
            pos += n;
            if (raw.empty()) break;
        }
        wr32be(z, (b2 << 16) | a);
        png_chunk(out, "IDAT", z);
        png_chunk(out, "IEND", {});
        return write_all(path, out);
    }
    // default: 24-bit BMP, bottom-up
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    out.push_back('B');
    out.push_back('M');
    wr32(out, (uint32_t)(54 + stride * h));
    wr32(out, 0);
    wr32(out, 54);
    wr32(out, 40);
    wr32(out, (uint32_t)w);
    wr32(out, (uint32_t)h);
    wr16(out, 1);
    wr16(out, 24);
    wr32(out, 0);
    wr32(out, (uint32_t)(stride * h));
    wr32(out, 2835);
    wr32(out, 2835);
    wr32(out, 0);
    wr32(out, 0);
    std::vector<unsigned char> row(stride, 0);
    for (int r = h - 1; r >= 0; --r) {
        std::memcpy(row.data(), img.ptr<unsigned char>(r), (size_t)w * 3);
        out.insert(out.end(), row.begin(), row.end());
    }
    return write_all(path, out);
}

}  // namespace nle
