// JPEG reader for the enhance / denoise CLIs: `cv::imread` (reference src/enhance.cpp:33) reads data/paper.jpg (progressive)
// and data/rock2.jpg (baseline) in the README's sample table (README.md:81-82).  Stand-in for OpenCV's imgcodecs, no filter
// arithmetic here.
//
// Scope: Huffman-coded 8-bit DCT JPEG, baseline / extended sequential (SOF0, SOF1) and progressive (SOF2: spectral
// selection and successive approximation), 1 (grey) or 3 components, sampling factors 1..2 per axis (4:4:4, 4:2:2, 4:4:0,
// 4:2:0), restart intervals, interleaved and non-interleaved scans.  Refused (empty image): arithmetic coding, 12-bit,
// lossless, hierarchical, CMYK / 4 components, other sampling factors.
//
// The input of the filter is the decoded image, and "a JPEG decoded by a different decoder is not the same input": the
// pixel pipeline therefore follows the de-facto reference decoder's DEFAULT arithmetic step by step, as published in the
// IJG / libjpeg documentation and the JPEG standard (ITU T.81) -- which is what OpenCV's imread runs:
//   * the "slow integer" inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants, two passes, 2 extra bits between),
//   * "fancy" chroma upsampling (triangle filter: 3/4 nearer + 1/4 farther sample, the published rounding pattern),
//   * YCbCr -> RGB in 16-bit fixed point.
// tests/test_jpeg.py holds the output to Pillow's (libjpeg-turbo, same defaults) bit for bit on files of every supported kind.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "nle/image_io.hpp"

namespace nle {
namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    int maxcode[18];        // largest code of each length (-1: none), [17] = sentinel
    int valptr[17];         // index of the first symbol of each length
    int mincode[17];
    unsigned char vals[256];
    short fast[512];        // 9-bit prefix -> (length << 8 | symbol), -1: longer code

    bool build(const unsigned char* counts, const unsigned char* symbols, int nsym) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            k += counts[len - 1];
            code += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;  // over-subscribed
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        if (k != nsym || nsym > 256) return false;
        std::memcpy(vals, symbols, (size_t)nsym);
        for (int i = 0; i < 512; ++i) fast[i] = -1;
        for (int len = 1; len <= 9; ++len)
            for (int i = 0; i < counts[len - 1]; ++i) {
                const int c = (mincode[len] + i) << (9 - len);
                for (int f = 0; f < (1 << (9 - len)); ++f) fast[c + f] = (short)((len << 8) | vals[valptr[len] + i]);
            }
        present = true;
        return true;
    }
};

// entropy-coded segment reader: FF00 -> FF, stops at any other marker (further bits read as zeros, `hit_marker` set)
struct Bits {
    const unsigned char* p;
    const unsigned char* end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    int overrun = 0;  // zero bytes supplied past the data: a sane stream needs only a few

    void fill() {
        while (n <= 24) {
            unsigned b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else {
                        hit_marker = true;
                        b = 0;
                    }
                } else {
                    ++p;
                }
            } else {
                hit_marker = true;
            }
            if (hit_marker) ++overrun;
            acc |= b << (24 - n);
            n += 8;
        }
    }
    int peek(int k) {
        if (n < k) fill();
        return (int)(acc >> (32 - k));
    }
    void skip(int k) {
        acc <<= k;
        n -= k;
    }
    int get(int k) {
        if (k == 0) return 0;
        const int v = peek(k);
        skip(k);
        return v;
    }
    int bit() { return get(1); }
    // after a restart interval: drop the partial byte, step over the RSTn marker if it is there
    void restart() {
        acc = 0;
        n = 0;
        hit_marker = false;
        overrun = 0;
        while (p + 1 < end && !(p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7)) {
            if (p[0] == 0xff && p[1] != 0x00 && p[1] != 0xff) return;  // some other marker: leave it
            ++p;
        }
        if (p + 1 < end) p += 2;
    }
};

inline int decode_symbol(Bits& br, const HuffTable& h) {
    const int look = br.peek(9);
    const int f = h.fast[look];
    if (f >= 0) {
        br.skip(f >> 8);
        return f & 255;
    }
    int code = br.peek(16), len = 10;
    for (; len <= 16; ++len) {
        const int c = code >> (16 - len);
        if (h.maxcode[len] >= 0 && c <= h.maxcode[len] && c >= h.mincode[len]) {
            br.skip(len);
            return h.vals[h.valptr[len] + c - h.mincode[len]];
        }
    }
    br.skip(16);
    return -1;  // invalid code
}

inline int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int bw = 0, bh = 0;      // blocks allocated (MCU padded)
    int cw = 0, ch = 0;      // downsampled size in samples: ceil(W h / hmax), ceil(H v / vmax)
    int td = 0, ta = 0;      // tables of the current scan
    int pred = 0;
    std::vector<short> coef;             // [bh][bw][64], natural order
    std::vector<unsigned char> plane;    // [bh * 8][bw * 8]
};

struct Decoder {
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    int adobe_transform = -1;
    unsigned short quant[4][64];
    bool have_quant[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[3];
    int eobrun = 0;
};

constexpr int kMaxDim = 65535;
constexpr size_t kMaxBlocks = (size_t)1 << 24;  // 2^24 blocks = 2^30 coefficients (2 GiB of shorts): images up to ~32k x 32k

bool parse_sof(Decoder& d, const unsigned char* s, size_t len, bool progressive, size_t file_bytes) {
    if (d.have_frame || len < 6) return false;
    if (s[0] != 8) return false;  // 8-bit samples only
    d.H = (s[1] << 8) | s[2];
    d.W = (s[3] << 8) | s[4];
    d.ncomp = s[5];
    if (d.W <= 0 || d.H <= 0 || d.W > kMaxDim || d.H > kMaxDim) return false;
    if ((d.ncomp != 1 && d.ncomp != 3) || len < 6 + 3 * (size_t)d.ncomp) return false;
    d.hmax = d.vmax = 1;
    for (int i = 0; i < d.ncomp; ++i) {
        Component& c = d.comp[i];
        c.id = s[6 + 3 * i];
        c.h = s[7 + 3 * i] >> 4;
        c.v = s[7 + 3 * i] & 15;
        c.tq = s[8 + 3 * i];
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return false;
        d.hmax = c.h > d.hmax ? c.h : d.hmax;
        d.vmax = c.v > d.vmax ? c.v : d.vmax;
    }
    if (d.ncomp == 1) d.comp[0].h = d.comp[0].v = d.hmax = d.vmax = 1;  // a single component is never subsampled
    else if (d.comp[1].h != 1 || d.comp[1].v != 1 || d.comp[2].h != 1 || d.comp[2].v != 1) return false;  // chroma at the coarse rate only
    d.mcux = (d.W + 8 * d.hmax - 1) / (8 * d.hmax);
    d.mcuy = (d.H + 8 * d.vmax - 1) / (8 * d.vmax);
    size_t blocks = 0;
    for (int i = 0; i < d.ncomp; ++i) {
        Component& c = d.comp[i];
        c.bw = d.mcux * c.h;
        c.bh = d.mcuy * c.v;
        c.cw = (d.W * c.h + d.hmax - 1) / d.hmax;
        c.ch = (d.H * c.v + d.vmax - 1) / d.vmax;
        blocks += (size_t)c.bw * c.bh;
    }
    if (blocks > kMaxBlocks) return false;
    // every coded block costs at least one bit of entropy data (its DC code), so a file of n bytes cannot hold more than
    // 8 n blocks: a 200-byte header claiming 32k x 32k is refused HERE, before gigabytes are allocated on its word
    if (blocks > 8 * file_bytes) return false;
    for (int i = 0; i < d.ncomp; ++i) d.comp[i].coef.assign((size_t)d.comp[i].bw * d.comp[i].bh * 64, 0);
    d.progressive = progressive;
    d.have_frame = true;
    return true;
}

bool parse_dqt(Decoder& d, const unsigned char* s, size_t len) {
    size_t i = 0;
    while (i < len) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        ++i;
        if (tq > 3 || pq > 1 || i + (pq ? 128 : 64) > len) return false;
        for (int k = 0; k < 64; ++k) {
            const unsigned v = pq ? (unsigned)((s[i] << 8) | s[i + 1]) : s[i];
            i += pq ? 2 : 1;
            d.quant[tq][kZigzag[k]] = (unsigned short)v;
        }
        d.have_quant[tq] = true;
    }
    return true;
}

bool parse_dht(Decoder& d, const unsigned char* s, size_t len) {
    size_t i = 0;
    while (i < len) {
        if (i + 17 > len) return false;
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (tc > 1 || th > 3) return false;
        int nsym = 0;
        for (int k = 0; k < 16; ++k) nsym += s[i + 1 + k];
        if (nsym > 256 || i + 17 + (size_t)nsym > len) return false;
        HuffTable& t = tc ? d.ac[th] : d.dc[th];
        if (!t.build(s + i + 1, s + i + 17, nsym)) return false;
        i += 17 + (size_t)nsym;
    }
    return true;
}

// ---- block decoders.  `blk`: 64 coefficients in natural order.  Return false on an invalid code.
bool block_baseline(Decoder& d, Bits& br, Component& c, short* blk) {
    const HuffTable &hd = d.dc[c.td], &ha = d.ac[c.ta];
    const int t = decode_symbol(br, hd);
    if (t < 0 || t > 15) return false;
    c.pred = (int)((unsigned)c.pred + (unsigned)extend(br.get(t), t));  // wraps on a corrupt stream instead of overflowing
    blk[0] = (short)c.pred;
    for (int k = 1; k < 64;) {
        const int rs = decode_symbol(br, ha);
        if (rs < 0) return false;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r != 15) break;
            k += 16;
        } else {
            k += r;
            if (k > 63) return false;
            blk[kZigzag[k]] = (short)extend(br.get(s), s);
            ++k;
        }
    }
    return true;
}

bool block_dc_first(Decoder& d, Bits& br, Component& c, short* blk, int al) {
    const int t = decode_symbol(br, d.dc[c.td]);
    if (t < 0 || t > 15) return false;
    c.pred = (int)((unsigned)c.pred + (unsigned)extend(br.get(t), t));
    blk[0] = (short)((unsigned)c.pred << al);
    return true;
}

void block_dc_refine(Bits& br, short* blk, int al) {
    if (br.bit()) blk[0] = (short)(blk[0] | (1 << al));
}

bool block_ac_first(Decoder& d, Bits& br, Component& c, short* blk, int ss, int se, int al) {
    if (d.eobrun > 0) {
        --d.eobrun;
        return true;
    }
    const HuffTable& ha = d.ac[c.ta];
    for (int k = ss; k <= se;) {
        const int rs = decode_symbol(br, ha);
        if (rs < 0) return false;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r < 15) {
                d.eobrun = (1 << r) - 1;
                if (r) d.eobrun += br.get(r);
                break;
            }
            k += 16;
        } else {
            k += r;
            if (k > se) return false;
            blk[kZigzag[k]] = (short)(extend(br.get(s), s) * (1 << al));
            ++k;
        }
    }
    return true;
}

bool block_ac_refine(Decoder& d, Bits& br, Component& c, short* blk, int ss, int se, int al) {
    const int p1 = 1 << al, m1 = -(1 << al);
    const HuffTable& ha = d.ac[c.ta];
    int k = ss;
    auto refine = [&](short& co) {  // one correction bit for a coefficient that is already non-zero
        if (br.bit() && (co & p1) == 0) co = (short)(co + (co >= 0 ? p1 : m1));
    };
    if (d.eobrun == 0) {
        while (k <= se) {
            const int rs = decode_symbol(br, ha);
            if (rs < 0) return false;
            int r = rs >> 4;
            const int s = rs & 15;
            int value = 0;
            if (s) {
                if (s != 1) return false;
                value = br.bit() ? p1 : m1;
            } else if (r != 15) {
                d.eobrun = 1 << r;  // this block included
                if (r) d.eobrun += br.get(r);
                break;
            }
            // skip r zero-history coefficients (refining the non-zero ones passed on the way), then place `value`
            while (k <= se) {
                short& co = blk[kZigzag[k]];
                if (co != 0) refine(co);
                else if (--r < 0) break;
                ++k;
            }
            if (s && k <= se) blk[kZigzag[k]] = (short)value;
            ++k;
        }
    }
    if (d.eobrun > 0) {
        for (; k <= se; ++k) {
            short& co = blk[kZigzag[k]];
            if (co != 0) refine(co);
        }
        --d.eobrun;
    }
    return true;
}

// one scan: header at s (length len), entropy data follows at *pos; advances *pos to the next marker
bool decode_scan(Decoder& d, const unsigned char* s, size_t len, const std::vector<unsigned char>& b, size_t* pos) {
    if (!d.have_frame || len < 1) return false;
    const int ns = s[0];
    if (ns < 1 || ns > d.ncomp || len < 1 + 2 * (size_t)ns + 3) return false;
    int ci[3];
    for (int i = 0; i < ns; ++i) {
        int k = 0;
        while (k < d.ncomp && d.comp[k].id != s[1 + 2 * i]) ++k;
        if (k == d.ncomp) return false;
        for (int j = 0; j < i; ++j)
            if (ci[j] == k) return false;
        ci[i] = k;
        d.comp[k].td = s[2 + 2 * i] >> 4;
        d.comp[k].ta = s[2 + 2 * i] & 15;
        if (d.comp[k].td > 3 || d.comp[k].ta > 3) return false;
    }
    const int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
    if (d.progressive) {
        if (ss > se || se > 63 || al > 13 || ah > 13) return false;
        if (ss == 0 && se != 0) return false;      // DC and AC never share a scan
        if (ss > 0 && ns != 1) return false;       // AC scans hold one component
    } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
        return false;
    }
    const bool need_dc = ss == 0 && ah == 0, need_ac = !d.progressive || ss > 0;
    for (int i = 0; i < ns; ++i) {
        const Component& c = d.comp[ci[i]];
        if (need_dc && !d.dc[c.td].present) return false;
        if (need_ac && !d.ac[c.ta].present) return false;
    }
    Bits br{b.data() + *pos, b.data() + b.size()};
    for (int i = 0; i < ns; ++i) d.comp[ci[i]].pred = 0;
    d.eobrun = 0;
    auto one_block = [&](Component& c, short* blk) -> bool {
        if (!d.progressive) return block_baseline(d, br, c, blk);
        if (ss == 0) {
            if (ah == 0) return block_dc_first(d, br, c, blk, al);
            block_dc_refine(br, blk, al);
            return true;
        }
        return ah == 0 ? block_ac_first(d, br, c, blk, ss, se, al) : block_ac_refine(d, br, c, blk, ss, se, al);
    };
    int units_x, units_y;  // MCUs of this scan
    if (ns == 1) {  // non-interleaved: the component's own blocks, ceil(size / 8)
        const Component& c = d.comp[ci[0]];
        units_x = (c.cw + 7) / 8;
        units_y = (c.ch + 7) / 8;
    } else {
        units_x = d.mcux;
        units_y = d.mcuy;
    }
    int until_restart = d.restart_interval;
    bool ok = true;
    for (int my = 0; my < units_y && ok; ++my)
        for (int mx = 0; mx < units_x && ok; ++mx) {
            if (d.restart_interval) {
                if (until_restart == 0) {
                    br.restart();
                    for (int i = 0; i < ns; ++i) d.comp[ci[i]].pred = 0;
                    d.eobrun = 0;
                    until_restart = d.restart_interval;
                }
                --until_restart;
            }
            if (ns == 1) {
                Component& c = d.comp[ci[0]];
                ok = one_block(c, &c.coef[((size_t)my * c.bw + mx) * 64]);
            } else {
                for (int i = 0; i < ns && ok; ++i) {
                    Component& c = d.comp[ci[i]];
                    for (int v = 0; v < c.v && ok; ++v)
                        for (int h = 0; h < c.h && ok; ++h)
                            ok = one_block(c, &c.coef[((size_t)(my * c.v + v) * c.bw + (mx * c.h + h)) * 64]);
                }
            }
            if (br.overrun > 64) ok = false;  // truncated stream: stop instead of inventing an image from zeros
        }
    // the next marker: the reader has looked ahead by at most a few bytes and never past a marker
    const unsigned char* q = br.p;
    const unsigned char* end = b.data() + b.size();
    while (q + 1 < end && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7))) ++q;
    *pos = (size_t)(q - b.data());
    return ok;
}

// ---- inverse DCT, "slow integer" form (Loeffler, Ligtenberg, Moschytz 1989; constants scaled by 2^13; the column pass
// keeps 2 extra bits).  in: 64 dequantised coefficients (natural order), out: 8 x 8 samples, stride `stride`
inline unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idct_block(const short* co, const unsigned short* q, unsigned char* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                  F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    // 64-bit intermediates: identical results on any valid stream (the 32-bit form never overflows there), no undefined
    // behaviour on a corrupt one
    typedef long long i64;
    i64 ws[64];
    auto butterfly = [&](i64 i0, i64 i1, i64 i2, i64 i3, i64 i4, i64 i5, i64 i6, i64 i7, i64 (&o)[8]) {
        // even part
        i64 z2 = i2, z3 = i6;
        i64 z1 = (z2 + z3) * F0_541;
        i64 tmp2 = z1 + z3 * (-F1_847);
        i64 tmp3 = z1 + z2 * F0_765;
        i64 tmp0 = (i0 + i4) * (1 << CB), tmp1 = (i0 - i4) * (1 << CB);
        const i64 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        // odd part
        tmp0 = i7;
        tmp1 = i5;
        tmp2 = i3;
        tmp3 = i1;
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        i64 z4 = tmp1 + tmp3;
        const i64 z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298;
        tmp1 *= F2_053;
        tmp2 *= F3_072;
        tmp3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        o[0] = tmp10 + tmp3;
        o[7] = tmp10 - tmp3;
        o[1] = tmp11 + tmp2;
        o[6] = tmp11 - tmp2;
        o[2] = tmp12 + tmp1;
        o[5] = tmp12 - tmp1;
        o[3] = tmp13 + tmp0;
        o[4] = tmp13 - tmp0;
    };
    auto descale = [](i64 x, int n) { return (x + ((i64)1 << (n - 1))) >> n; };
    for (int c = 0; c < 8; ++c) {  // columns
        i64 in[8];
        for (int r = 0; r < 8; ++r) in[r] = (i64)co[r * 8 + c] * (i64)q[r * 8 + c];
        i64 o[8];
        butterfly(in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], o);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = descale(o[r], CB - P1);
    }
    for (int r = 0; r < 8; ++r) {  // rows
        const i64* w = &ws[r * 8];
        i64 o[8];
        butterfly(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], o);
        for (int c = 0; c < 8; ++c) {
            const i64 v = descale(o[c], CB + P1 + 3) + 128;
            out[r * stride + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// ---- chroma upsampling to full resolution ("fancy": triangle filter).  src: cw x ch samples (stride sstride, rows past
// ch are not read: the last row stands in), dst: (cw * hx) x (ch * vx), only the W x H corner is kept by the caller
void upsample(const Component& c, int hx, int vx, std::vector<unsigned char>& dst, int* dst_stride) {
    const int cw = c.cw, ch = c.ch, ss = c.bw * 8;
    const unsigned char* src = c.plane.data();
    const int ow = cw * hx, oh = ch * vx;
    *dst_stride = ow;
    dst.assign((size_t)ow * oh, 0);
    auto row = [&](int r) { return src + (size_t)(r < 0 ? 0 : (r >= ch ? ch - 1 : r)) * ss; };
    if (hx == 1 && vx == 1) {
        for (int r = 0; r < ch; ++r) std::memcpy(&dst[(size_t)r * ow], row(r), (size_t)cw);
        return;
    }
    const bool fancy = cw > 2;  // narrower components are replicated
    if (!fancy) {
        for (int r = 0; r < oh; ++r)
            for (int x = 0; x < ow; ++x) dst[(size_t)r * ow + x] = row(r / vx)[x / hx];
        return;
    }
    if (hx == 2 && vx == 1) {
        for (int r = 0; r < ch; ++r) {
            const unsigned char* in = row(r);
            unsigned char* o = &dst[(size_t)r * ow];
            o[0] = in[0];
            o[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
            for (int i = 1; i < cw - 1; ++i) {
                o[2 * i] = (unsigned char)((in[i] * 3 + in[i - 1] + 1) >> 2);
                o[2 * i + 1] = (unsigned char)((in[i] * 3 + in[i + 1] + 2) >> 2);
            }
            o[2 * cw - 2] = (unsigned char)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
            o[2 * cw - 1] = in[cw - 1];
        }
        return;
    }
    if (hx == 1 && vx == 2) {
        for (int r = 0; r < oh; ++r) {
            const unsigned char* near = row(r >> 1);
            const unsigned char* far = row((r & 1) ? (r >> 1) + 1 : (r >> 1) - 1);
            const int bias = (r & 1) ? 2 : 1;
            unsigned char* o = &dst[(size_t)r * ow];
            for (int x = 0; x < cw; ++x) o[x] = (unsigned char)((near[x] * 3 + far[x] + bias) >> 2);
        }
        return;
    }
    // hx == 2 && vx == 2
    for (int r = 0; r < oh; ++r) {
        const unsigned char* near = row(r >> 1);
        const unsigned char* far = row((r & 1) ? (r >> 1) + 1 : (r >> 1) - 1);
        unsigned char* o = &dst[(size_t)r * ow];
        int thiss = near[0] * 3 + far[0], nexts = near[1] * 3 + far[1], lasts;
        o[0] = (unsigned char)((thiss * 4 + 8) >> 4);
        o[1] = (unsigned char)((thiss * 3 + nexts + 7) >> 4);
        lasts = thiss;
        thiss = nexts;
        for (int i = 1; i < cw - 1; ++i) {
            nexts = near[i + 1] * 3 + far[i + 1];
            o[2 * i] = (unsigned char)((thiss * 3 + lasts + 8) >> 4);
            o[2 * i + 1] = (unsigned char)((thiss * 3 + nexts + 7) >> 4);
            lasts = thiss;
            thiss = nexts;
        }
        o[2 * cw - 2] = (unsigned char)((thiss * 3 + lasts + 8) >> 4);
        o[2 * cw - 1] = (unsigned char)((thiss * 4 + 7) >> 4);
    }
}

}  // namespace

Image read_jpeg(const std::vector<unsigned char>& b) {
    if (b.size() < 4 || b[0] != 0xff || b[1] != 0xd8) return Image();
    Decoder d;
    size_t pos = 2;
    bool seen_eoi = false, any_scan = false;
    bool rgb_ids = false;
    while (pos + 1 < b.size() && !seen_eoi) {
        if (b[pos] != 0xff) {  // garbage between segments: resynchronise on the next 0xff
            ++pos;
            continue;
        }
        const unsigned m = b[pos + 1];
        if (m == 0xff) {  // fill byte
            ++pos;
            continue;
        }
        pos += 2;
        if (m == 0xd9) {
            seen_eoi = true;
            break;
        }
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7) || m == 0x00) continue;  // stand-alone markers
        if (pos + 2 > b.size()) break;
        const size_t seglen = ((size_t)b[pos] << 8) | b[pos + 1];
        if (seglen < 2 || pos + seglen > b.size()) return Image();
        const unsigned char* s = &b[pos + 2];
        const size_t len = seglen - 2;
        pos += seglen;
        switch (m) {
            case 0xc0:
            case 0xc1:
                if (!parse_sof(d, s, len, false, b.size())) return Image();
                break;
            case 0xc2:
                if (!parse_sof(d, s, len, true, b.size())) return Image();
                break;
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                return Image();  // lossless, hierarchical, arithmetic coding
            case 0xc4:
                if (!parse_dht(d, s, len)) return Image();
                break;
            case 0xdb:
                if (!parse_dqt(d, s, len)) return Image();
                break;
            case 0xdd:
                if (len < 2) return Image();
                d.restart_interval = (s[0] << 8) | s[1];
                break;
            case 0xee:  // Adobe: the colour transform flag
                if (len >= 12 && std::memcmp(s, "Adobe", 5) == 0) d.adobe_transform = s[11];
                break;
            case 0xda:
                if (!decode_scan(d, s, len, b, &pos)) {
                    if (!any_scan) return Image();
                    seen_eoi = true;  // a damaged later scan: keep what the earlier ones gave (as the usual decoders do)
                }
                any_scan = true;
                break;
            default:
                break;  // APPn, COM, ...
        }
    }
    if (!d.have_frame || !any_scan) return Image();
    for (int i = 0; i < d.ncomp; ++i)
        if (!d.have_quant[d.comp[i].tq]) return Image();
    rgb_ids = d.ncomp == 3 && d.comp[0].id == 'R' && d.comp[1].id == 'G' && d.comp[2].id == 'B';
    const bool ycc = d.ncomp == 3 && (d.adobe_transform >= 0 ? d.adobe_transform == 1 : !rgb_ids);
    // dequantise + inverse DCT, component planes
    for (int i = 0; i < d.ncomp; ++i) {
        Component& c = d.comp[i];
        const int stride = c.bw * 8;
        c.plane.assign((size_t)stride * c.bh * 8, 0);
        for (int by = 0; by < c.bh; ++by)
            for (int bx = 0; bx < c.bw; ++bx)
                idct_block(&c.coef[((size_t)by * c.bw + bx) * 64], d.quant[c.tq], &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
        c.coef.clear();
        c.coef.shrink_to_fit();
    }
    Image img(d.H, d.W, NLE_8U, 3);
    if (d.ncomp == 1) {
        const Component& c = d.comp[0];
        for (int r = 0; r < d.H; ++r) {
            const unsigned char* y = &c.plane[(size_t)r * c.bw * 8];
            unsigned char* o = img.ptr<unsigned char>(r);
            for (int x = 0; x < d.W; ++x) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = y[x];
        }
        return img;
    }
    std::vector<unsigned char> up[3];
    int ust[3];
    for (int i = 0; i < 3; ++i) upsample(d.comp[i], d.hmax / d.comp[i].h, d.vmax / d.comp[i].v, up[i], &ust[i]);
    // YCbCr -> RGB in 16-bit fixed point: R = Y + 1.402 Cr', G = Y - 0.34414 Cb' - 0.71414 Cr', B = Y + 1.772 Cb'
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
        const int x = i - 128;
        cr_r[i] = (91881 * x + 32768) >> 16;
        cb_b[i] = (116130 * x + 32768) >> 16;
        cr_g[i] = -46802 * x;
        cb_g[i] = -22554 * x + 32768;
    }
    for (int r = 0; r < d.H; ++r) {
        const unsigned char* p0 = &up[0][(size_t)r * ust[0]];
        const unsigned char* p1 = &up[1][(size_t)r * ust[1]];
        const unsigned char* p2 = &up[2][(size_t)r * ust[2]];
        unsigned char* o = img.ptr<unsigned char>(r);
        for (int x = 0; x < d.W; ++x) {
            if (ycc) {
                const int y = p0[x], cb = p1[x], cr = p2[x];
                o[3 * x + 2] = clamp8(y + cr_r[cr]);
                o[3 * x + 1] = clamp8(y + ((cb_g[cb] + cr_g[cr]) >> 16));
                o[3 * x + 0] = clamp8(y + cb_b[cb]);
            } else {
                o[3 * x + 2] = p0[x];
                o[3 * x + 1] = p1[x];
                o[3 * x + 0] = p2[x];
            }
        }
    }
    return img;
}

// ------------------------------------------------------------------------------------------------ writer
// Baseline JPEG out (cv::imwrite's role for a .jpg result, reference src/enhance.cpp:47): 8-bit YCbCr 4:2:0, the standard
// (ITU T.81 Annex K) quantisation tables scaled by `quality` the IJG way and the standard Huffman tables, JFIF header.
// Colour conversion and 2 x 2 chroma averaging in the integer arithmetic the IJG documentation gives; the forward DCT in
// double precision.  An output format, not part of any parity claim.
namespace {

const unsigned char kQLum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                 69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                                 81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72,  92,  95,  98, 112, 100, 103, 99};
const unsigned char kQChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                 99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const unsigned char kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const unsigned char kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const unsigned char kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const unsigned char kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const unsigned char kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const unsigned char kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const unsigned char kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable {
    unsigned short code[256];
    unsigned char len[256];
    void build(const unsigned char* bits, const unsigned char* vals) {
        std::memset(len, 0, sizeof len);
        int code_ = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            for (int i = 0; i < bits[l - 1]; ++i, ++k) {
                code[vals[k]] = (unsigned short)code_++;
                len[vals[k]] = (unsigned char)l;
            }
            code_ <<= 1;
        }
    }
};

struct BitSink {
    std::vector<unsigned char>& out;
    uint32_t acc = 0;
    int n = 0;
    void put(unsigned v, int bits) {  // bits <= 16
        acc = (acc << bits) | (v & ((1u << bits) - 1u));
        n += bits;
        while (n >= 8) {
            const unsigned char byte = (unsigned char)(acc >> (n - 8));
            out.push_back(byte);
            if (byte == 0xff) out.push_back(0x00);
            n -= 8;
        }
    }
    void flush() {
        if (n > 0) put(0x7f, 8 - n);  // pad with ones
    }
};

void put16(std::vector<unsigned char>& o, unsigned v) {
    o.push_back((unsigned char)(v >> 8));
    o.push_back((unsigned char)v);
}

void encode_block(BitSink& bs, const double* px /* 64 level-shifted samples */, const unsigned short* q, int* pred,
                  const EncTable& dc, const EncTable& ac) {
    // forward DCT, separable, double precision
    static double C[8][8];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 8; ++u)
            for (int x = 0; x < 8; ++x) C[u][x] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    double t[64], f[64];
    for (int y = 0; y < 8; ++y)
        for (int u = 0; u < 8; ++u) {
            double s = 0.0;
            for (int x = 0; x < 8; ++x) s += C[u][x] * px[y * 8 + x];
            t[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) {
            double s = 0.0;
            for (int y = 0; y < 8; ++y) s += C[v][y] * t[y * 8 + u];
            f[v * 8 + u] = s;
        }
    int zz[64];
    for (int k = 0; k < 64; ++k) {
        const int nat = kZigzag[k];
        zz[k] = (int)std::lrint(f[nat] / (double)q[nat]);
    }
    auto category = [](int v) {
        int a = v < 0 ? -v : v, c = 0;
        while (a) {
            ++c;
            a >>= 1;
        }
        return c;
    };
    const int diff = zz[0] - *pred;
    *pred = zz[0];
    int cat = category(diff);
    bs.put(dc.code[cat], dc.len[cat]);
    if (cat) bs.put((unsigned)(diff < 0 ? diff - 1 : diff), cat);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        if (zz[k] == 0) {
            ++run;
            continue;
        }
        while (run > 15) {
            bs.put(ac.code[0xf0], ac.len[0xf0]);
            run -= 16;
        }
        cat = category(zz[k]);
        const int sym = (run << 4) | cat;
        bs.put(ac.code[sym], ac.len[sym]);
        bs.put((unsigned)(zz[k] < 0 ? zz[k] - 1 : zz[k]), cat);
        run = 0;
    }
    if (run > 0) bs.put(ac.code[0x00], ac.len[0x00]);
}

}  // namespace

bool write_jpeg(std::vector<unsigned char>* out_bytes, const Image& img, int quality) {
    if (img.empty() || img.channels() != 3 || img.depth() != NLE_8U || img.rows > kMaxDim || img.cols > kMaxDim) return false;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    unsigned short ql[64], qc[64];
    for (int i = 0; i < 64; ++i) {
        int a = ((int)kQLum[i] * scale + 50) / 100, b = ((int)kQChr[i] * scale + 50) / 100;
        ql[i] = (unsigned short)(a < 1 ? 1 : (a > 255 ? 255 : a));
        qc[i] = (unsigned short)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    const int H = img.rows, W = img.cols, mx = (W + 15) / 16, my = (H + 15) / 16, PW = mx * 16, PH = my * 16;
    // YCbCr planes padded to whole MCUs by edge replication; chroma averaged 2 x 2 (bias 1, 2, 1, 2, ...)
    std::vector<unsigned char> Y((size_t)PW * PH), Cb((size_t)PW * PH), Cr((size_t)PW * PH);
    for (int r = 0; r < PH; ++r) {
        const unsigned char* src = img.ptr<unsigned char>(r < H ? r : H - 1);
        for (int c = 0; c < PW; ++c) {
            const int cc = c < W ? c : W - 1;
            const int B = src[3 * cc], G = src[3 * cc + 1], R = src[3 * cc + 2];
            Y[(size_t)r * PW + c] = (unsigned char)((19595 * R + 38470 * G + 7471 * B + 32768) >> 16);
            Cb[(size_t)r * PW + c] = (unsigned char)((-11059 * R - 21709 * G + 32768 * B + (128 << 16) + 32767) >> 16);
            Cr[(size_t)r * PW + c] = (unsigned char)((32768 * R - 27439 * G - 5329 * B + (128 << 16) + 32767) >> 16);
        }
    }
    const int CW = PW / 2, CH = PH / 2;
    std::vector<unsigned char> cb2((size_t)CW * CH), cr2((size_t)CW * CH);
    for (int r = 0; r < CH; ++r)
        for (int c = 0; c < CW; ++c) {
            const size_t i0 = (size_t)(2 * r) * PW + 2 * c, i1 = i0 + PW;
            const int bias = 1 + (c & 1);
            cb2[(size_t)r * CW + c] = (unsigned char)((Cb[i0] + Cb[i0 + 1] + Cb[i1] + Cb[i1 + 1] + bias) >> 2);
            cr2[(size_t)r * CW + c] = (unsigned char)((Cr[i0] + Cr[i0 + 1] + Cr[i1] + Cr[i1 + 1] + bias) >> 2);
        }
    std::vector<unsigned char>& o = *out_bytes;
    o.clear();
    o.insert(o.end(), {0xff, 0xd8, 0xff, 0xe0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0x00, 0x01, 0x01, 0x00, 0x00, 0x01, 0x00, 0x01, 0x00, 0x00});
    for (int t = 0; t < 2; ++t) {  // DQT
        o.insert(o.end(), {0xff, 0xdb, 0x00, 0x43, (unsigned char)t});
        const unsigned short* q = t ? qc : ql;
        for (int k = 0; k < 64; ++k) o.push_back((unsigned char)q[kZigzag[k]]);
    }
    o.insert(o.end(), {0xff, 0xc0, 0x00, 0x11, 0x08});
    put16(o, (unsigned)H);
    put16(o, (unsigned)W);
    o.insert(o.end(), {0x03, 0x01, 0x22, 0x00, 0x02, 0x11, 0x01, 0x03, 0x11, 0x01});
    auto dht = [&](int cls, int id, const unsigned char* bits, const unsigned char* vals, int nv) {
        o.insert(o.end(), {0xff, 0xc4});
        put16(o, (unsigned)(2 + 1 + 16 + nv));
        o.push_back((unsigned char)((cls << 4) | id));
        o.insert(o.end(), bits, bits + 16);
        o.insert(o.end(), vals, vals + nv);
    };
    dht(0, 0, kDcLumBits, kDcVals, 12);
    dht(1, 0, kAcLumBits, kAcLumVals, 162);
    dht(0, 1, kDcChrBits, kDcVals, 12);
    dht(1, 1, kAcChrBits, kAcChrVals, 162);
    o.insert(o.end(), {0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00});
    EncTable dcl, acl, dcc, acc_;
    dcl.build(kDcLumBits, kDcVals);
    acl.build(kAcLumBits, kAcLumVals);
    dcc.build(kDcChrBits, kDcVals);
    acc_.build(kAcChrBits, kAcChrVals);
    BitSink bs{o};
    int py = 0, pcb = 0, pcr = 0;
    double blk[64];
    auto take = [&](const std::vector<unsigned char>& plane, int stride, int r0, int c0) {
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) blk[y * 8 + x] = (double)plane[(size_t)(r0 + y) * stride + c0 + x] - 128.0;
    };
    for (int m = 0; m < my; ++m)
        for (int n = 0; n < mx; ++n) {
            for (int v = 0; v < 2; ++v)
                for (int h = 0; h < 2; ++h) {
                    take(Y, PW, m * 16 + v * 8, n * 16 + h * 8);
                    encode_block(bs, blk, ql, &py, dcl, acl);
                }
            take(cb2, CW, m * 8, n * 8);
            encode_block(bs, blk, qc, &pcb, dcc, acc_);
            take(cr2, CW, m * 8, n * 8);
            encode_block(bs, blk, qc, &pcr, dcc, acc_);
        }
    bs.flush();
    o.insert(o.end(), {0xff, 0xd9});
    return true;
}

}  // namespace nle
