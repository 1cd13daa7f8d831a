// OpenCV's 8-bit BGR <-> Lab as integer table algorithms (imgproc color_lab.cpp: `RGB2Lab_b`, `Lab2RGBinteger`), the
// arithmetic behind cv::cvtColor in the reference's colour wrapper (src/filter.cpp:423, 440, 463).  OpenCV is a third-party
// dependency absent from the reference tree; what is restated here is its published algorithm, validated against the
// reference's own README output files (tests/test_oracle_readme_pairs.py: flower and brickwall byte for byte).
// Shared by the table builder (lab8_tables.cpp, host), the device kernels (colour.hip) and the C ABI (abi_ctx.hip).
#pragma once

#ifdef __HIPCC__
#define NLE_LAB8_HD __host__ __device__ inline
#else
#define NLE_LAB8_HD inline
#endif

namespace nlelab8 {

constexpr int kGammaN = 256;                 // sRGB decode, scaled by 255 * 8
constexpr int kCbrtN = 256 * 3 / 2 * 8;      // f(t) of L*a*b*, t = i / 2040, scaled by 2^15
constexpr int kBase = 1 << 14;               // Lab -> BGR: everything scaled by 2^14
constexpr int kYfN = 256 * 2;                // LabToYF: (y, fy) per 8-bit L
constexpr int kMinAB = -8145;                // smallest fx / fz that occurs
constexpr int kAbN = kBase * 9 / 4;          // abToXZ as a table (the ABI exports it; the kernels compute it)
constexpr int kInvGammaN = 1 << 12;          // sRGB encode of i / 4096, scaled by 255

// byte layout of the one device blob: gamma | cbrt (u16) | coeffs[9] (int) | yf (u16) | inv_gamma (u16) | inv_coeffs[9] (int)
constexpr int kOffGamma = 0;
constexpr int kOffCbrt = kOffGamma + 2 * kGammaN;
constexpr int kOffCoeffs = kOffCbrt + 2 * kCbrtN;
constexpr int kOffYf = kOffCoeffs + 4 * 9;
constexpr int kOffInvGamma = kOffYf + 2 * kYfN;
constexpr int kOffInvCoeffs = kOffInvGamma + 2 * kInvGammaN;
constexpr int kBlobBytes = kOffInvCoeffs + 4 * 9;
static_assert(kOffCoeffs % 4 == 0 && kOffInvCoeffs % 4 == 0, "int tables 4-byte aligned");

// fx, fz -> x, z (OpenCV's abToXZ_b): the two branches of the inverse of f(t), C integer division (toward zero)
NLE_LAB8_HD int ab_to_xz(int t) {
    return t <= 3390 ? t * 108 / 841 - kBase * 16 / 116 * 108 / 841 : t * t / kBase * t / kBase;
}
// fy + a / 500 and fy - b / 200 in units of 2^-14, from the 8-bit a and b
NLE_LAB8_HD int fx_of(int fy, int a8) { return fy + (((5 * a8 * 53687 + (1 << 7)) >> 13) - 128 * kBase / 500); }
NLE_LAB8_HD int fz_of(int fy, int b8) { return fy - (((b8 * 41943 + (1 << 4)) >> 9) - 128 * kBase / 200 + 1); }

// host side (lab8_tables.cpp)
void forward_tables(unsigned short* gamma, unsigned short* cbrt_tab, int* coeffs);
void inverse_tables(unsigned short* yf, unsigned short* inv_gamma, int* coeffs);

}  // namespace nlelab8
