// Image file I/O for the enhance CLI (stand-in for OpenCV highgui/imgcodecs, which the reference
// uses at src/enhance.cpp:33,47).  No arithmetic of the filter lives here.
#include "nle/image_io.hpp"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace nle {

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
    if (w <= 0 || hraw == 0 || (bpp != 24 && bpp != 32) || (comp != 0 && comp != 3)) return Image();
    const int h = hraw < 0 ? -hraw : hraw;
    const size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
    if (off + stride * h > b.size()) return Image();
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
    if (w <= 0 || h <= 0 || mx != 255) return Image();
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
