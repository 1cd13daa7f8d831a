// Tables of OpenCV's 8-bit BGR <-> Lab (lab8_fixed.h).  OpenCV builds them with its bit-exact soft-float in SINGLE
// precision; IEEE float / double arithmetic in the same order gives the same integers (this file is compiled with
// -ffp-contract=off).  Two details decide entries: the cube root is a quartic rational polynomial evaluated in double whose
// quotient is CUT to 24 bits (entries 49 and 628 of the f(t) table differ by one from a correctly rounded cube root), and
// the products with the table scales are single-precision products rounded half to even.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "lab8_fixed.h"

namespace nlelab8 {
namespace {

float cube_root_f32(float x) {  // x > 0
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) == 0) return 0.f;
    int ex = (int)((u >> 23) & 0xff) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;  // exponent of the cube root
    const uint64_t fb0 = ((uint64_t)(shx + 1023) << 52) | ((uint64_t)(u & 0x7fffffu) << 29);
    double fr;
    std::memcpy(&fr, &fb0, 8);  // 1/8 <= fr < 1
    const double num = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr + 119.1654824285581628956914143) * fr +
                         13.43250139086239872172837314) * fr + 0.1636161226585754240958355063);
    const double den = ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr + 168.5254414101568283957668343) * fr +
                         33.9905941350215598754191872) * fr + 1.0);
    fr = num / den;
    uint64_t fb;
    std::memcpy(&fb, &fr, 8);
    const int e = (int)((fb >> 52) & 0x7ff) - 1023 + ex + 127;
    const uint32_t out = ((uint32_t)e << 23) | (uint32_t)((fb >> 29) & 0x7fffffu);
    float y;
    std::memcpy(&y, &out, 4);
    return y;
}

const double kXn = 0.950456, kZn = 1.088754;

}  // namespace

void forward_tables(unsigned short* gamma, unsigned short* cbrt_tab, int* coeffs) {
    for (int i = 0; i < kGammaN; ++i) {
        const float x = (float)i / 255.f;
        const double xd = x;
        const float v = (float)(xd <= 0.04045 ? xd / 12.92 : std::pow((xd + 0.055) / (1.0 + 0.055), 2.4));
        gamma[i] = (unsigned short)std::nearbyintf(2040.f * v);
    }
    const float scale = 1.f / (255.f * 8.f);
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    for (int i = 0; i < kCbrtN; ++i) {
        const float x = scale * (float)i;
        const float v = x < lthresh ? std::fmaf(x, lscale, lbias) : cube_root_f32(x);
        cbrt_tab[i] = (unsigned short)std::nearbyintf(32768.f * v);
    }
    const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double wp[3] = {kXn, 1.0, kZn};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) coeffs[3 * r + c] = (int)std::nearbyint(4096.0 * M[r][c] / wp[r]);
}

void inverse_tables(unsigned short* yf, unsigned short* inv_gamma, int* coeffs) {
    const float B = (float)kBase;
    for (int i = 0; i < 256; ++i) {
        int y, fy;
        if (i <= 20) {  // 8 * 255 / 100: the linear piece of L*
            y = (int)std::nearbyintf((float)(i * kBase * 20 * 9) / (float)(17 * 29 * 29 * 29));
            const float t = 16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29);
            fy = (int)std::nearbyintf(B * t);
        } else {
            const float q0 = (float)(i * 100 * kBase) / (float)(255 * 116);
            const float q1 = (float)(16 * kBase) / 116.f;
            const float f = q0 + q1;
            fy = (int)std::nearbyintf(f);
            const float f2 = f * f;
            const float f3 = f2 * f;
            y = (int)std::nearbyintf(f3 / (float)((long long)kBase * kBase));
        }
        yf[2 * i] = (unsigned short)y;
        yf[2 * i + 1] = (unsigned short)fy;
    }
    const float inv_scale = 1.f / (float)kInvGammaN;
    for (int i = 0; i < kInvGammaN; ++i) {
        const double xd = inv_scale * (float)i;
        const float v = (float)(xd <= 0.0031308 ? xd * 12.92 : std::pow(xd, 1.0 / 2.4) * (1.0 + 0.055) - 0.055);
        inv_gamma[i] = (unsigned short)std::nearbyintf(255.f * v);
    }
    const double Mi[3][3] = {{3.240479, -1.53715, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
    const double wp[3] = {kXn, 1.0, kZn};
    for (int r = 0; r < 3; ++r)  // rows R, G, B; columns x, y, z
        for (int c = 0; c < 3; ++c) coeffs[3 * r + c] = (int)std::nearbyint(4096.0 * Mi[r][c] * wp[c]);
}

}  // namespace nlelab8
