// 8-bit BGR <-> Lab on the device: the colour wrapper either side of the hot path
// (reference src/filter.cpp:460-469 getLuminanceChannel, :422-426 and :434-440 in NLEFilter::enhance,
// where it is cv::cvtColor on 8-bit images).  Both directions are OpenCV's integer table algorithms (lab8_fixed.h: exact
// integers, the same as the host restatement in host/filter.cpp and as the oracle; the author's output files of the
// reference's README are reproduced byte for byte where the filtered L plane has no rounding tie).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "lab8_fixed.h"

namespace nlek {

namespace {
__device__ __forceinline__ int sat8i(float v) {
    return (int)fminf(255.f, fmaxf(0.f, rintf(v)));  // round half to even, saturate (cv convertTo CV_8U)
}
}  // namespace

// OpenCV's fixed-point 8-bit path (imgproc RGB2Lab_b, restated in include/nle.h at nle_lab8_tables): tables in LDS, integer
// arithmetic throughout -- bit-identical to the host restatement and to the oracle.  lut = gamma[256] | cbrt[3072] (u16)
// | coeffs[9] (int) as abi_ctx.hip uploads them; d_L (optional) = L channel as fp32
__global__ __launch_bounds__(256) void k_bgr2lab8(const unsigned char* __restrict__ bgr, long long n,
                                                  const double* __restrict__ lut, unsigned char* __restrict__ lab,
                                                  float* __restrict__ Lf) {
    constexpr int NCB = nlelab8::kCbrtN;
    __shared__ unsigned short sg[256], sc[NCB];
    __shared__ int sk[9];
    const unsigned char* blob = reinterpret_cast<const unsigned char*>(lut);
    const unsigned short* t16 = reinterpret_cast<const unsigned short*>(blob + nlelab8::kOffGamma);
    for (int i = threadIdx.x; i < 256; i += 256) sg[i] = t16[i];
    for (int i = threadIdx.x; i < NCB; i += 256) sc[i] = t16[256 + i];
    if (threadIdx.x < 9) sk[threadIdx.x] = reinterpret_cast<const int*>(blob + nlelab8::kOffCoeffs)[threadIdx.x];
    __syncthreads();
    auto sat = [](int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int B = sg[bgr[3 * i + 0]], G = sg[bgr[3 * i + 1]], R = sg[bgr[3 * i + 2]];
        const int fX = sc[(R * sk[0] + G * sk[1] + B * sk[2] + 2048) >> 12];
        const int fY = sc[(R * sk[3] + G * sk[4] + B * sk[5] + 2048) >> 12];
        const int fZ = sc[(R * sk[6] + G * sk[7] + B * sk[8] + 2048) >> 12];
        const unsigned char L8 = sat((296 * fY - 1336934 + 16384) >> 15);
        if (lab != nullptr) {
            lab[3 * i + 0] = L8;
            lab[3 * i + 1] = sat((500 * (fX - fY) + 128 * 32768 + 16384) >> 15);
            lab[3 * i + 2] = sat((200 * (fY - fZ) + 128 * 32768 + 16384) >> 15);
        }
        if (Lf != nullptr) Lf[i] = (float)L8;
    }
}

// L from Lf when given (clamped to [0,255] and rounded half-to-even like :434-436), else lab's own
// (the denoise wrapper also replaces a and b by filtered planes, clamped and rounded the same way, :391-399).
// OpenCV's Lab2RGBinteger: (y, fy) from a 256-entry table, fx / fz by fixed-point divisions, the inverse of f(t) in
// integers (ab_to_xz, computed rather than looked up: its table would be 147 KB), the matrix in 12-bit fixed point and a
// 4096-entry sRGB encoding table -- the two small tables in LDS.
__global__ __launch_bounds__(256) void k_lab2bgr8(const unsigned char* __restrict__ lab, const float* __restrict__ Lf,
                                                  const float* __restrict__ af, const float* __restrict__ bf,
                                                  long long n, const double* __restrict__ lut,
                                                  unsigned char* __restrict__ bgr) {
    using namespace nlelab8;
    __shared__ unsigned short syf[kYfN], sig[kInvGammaN];
    __shared__ int sk[9];
    const unsigned char* blob = reinterpret_cast<const unsigned char*>(lut);
    const unsigned short* gyf = reinterpret_cast<const unsigned short*>(blob + kOffYf);
    const unsigned short* gig = reinterpret_cast<const unsigned short*>(blob + kOffInvGamma);
    for (int i = threadIdx.x; i < kYfN; i += 256) syf[i] = gyf[i];
    for (int i = threadIdx.x; i < kInvGammaN; i += 256) sig[i] = gig[i];
    if (threadIdx.x < 9) sk[threadIdx.x] = reinterpret_cast<const int*>(blob + kOffInvCoeffs)[threadIdx.x];
    __syncthreads();
    auto enc = [&](int c0, int c1, int c2, int x, int y, int z) {
        const int v = (c0 * x + c1 * y + c2 * z + (1 << 13)) >> 14;
        return (unsigned char)sig[v < 0 ? 0 : (v > kInvGammaN - 1 ? kInvGammaN - 1 : v)];
    };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int L8 = Lf != nullptr ? sat8i(Lf[i]) : (int)lab[3 * i + 0];
        const int a8 = af != nullptr ? sat8i(af[i]) : (int)lab[3 * i + 1];
        const int b8 = bf != nullptr ? sat8i(bf[i]) : (int)lab[3 * i + 2];
        const int y = syf[2 * L8], fy = syf[2 * L8 + 1];
        const int x = ab_to_xz(fx_of(fy, a8)), z = ab_to_xz(fz_of(fy, b8));
        bgr[3 * i + 0] = enc(sk[6], sk[7], sk[8], x, y, z);
        bgr[3 * i + 1] = enc(sk[3], sk[4], sk[5], x, y, z);
        bgr[3 * i + 2] = enc(sk[0], sk[1], sk[2], x, y, z);
    }
}

hipError_t bgr2lab8(hipStream_t s, const unsigned char* d_bgr, long long n, const double* d_lut, unsigned char* d_lab,
                    float* d_L) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_bgr2lab8, dim3(grid), dim3(256), 0, s, d_bgr, n, d_lut, d_lab, d_L);
    return hipGetLastError();
}

hipError_t lab2bgr8(hipStream_t s, const unsigned char* d_lab, const float* d_L, const float* d_a, const float* d_b,
                    long long n, const double* d_lut, unsigned char* d_bgr) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_lab2bgr8, dim3(grid), dim3(256), 0, s, d_lab, d_L, d_a, d_b, n, d_lut, d_bgr);
    return hipGetLastError();
}

// split + convertTo(float) of one channel of an interleaved 8-bit 3-channel image (src/filter.cpp:363,376-378)
__global__ __launch_bounds__(256) void k_channel8(const unsigned char* __restrict__ img, long long n, int ch,
                                                  float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)img[3 * i + ch];
}

hipError_t channel8(hipStream_t s, const unsigned char* d_img, long long n, int ch, float* d_out) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_channel8, dim3(grid), dim3(256), 0, s, d_img, n, ch, d_out);
    return hipGetLastError();
}

// the tail of NLEFilter::enhance on the filtered plane alone (src/filter.cpp:434-436): clamp to [0, 255], convertTo(CV_8U)
// (round half to even).  Four pixels per thread where the plane allows 16-byte loads.
__global__ __launch_bounds__(256) void k_plane_to_u8(const float* __restrict__ y, long long n, unsigned char* __restrict__ out) {
    const long long n4 = n >> 2;
    const bool vec = ((reinterpret_cast<size_t>(y) & 15) | (reinterpret_cast<size_t>(out) & 3)) == 0;
    if (vec) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4*>(y)[i];
            const unsigned r = (unsigned)sat8i(v.x) | ((unsigned)sat8i(v.y) << 8) | ((unsigned)sat8i(v.z) << 16) | ((unsigned)sat8i(v.w) << 24);
            reinterpret_cast<unsigned*>(out)[i] = r;
        }
    }
    for (long long i = (vec ? n4 * 4 : 0) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (unsigned char)sat8i(y[i]);
}

hipError_t plane_to_u8(hipStream_t s, const float* d_y, long long n, unsigned char* d_out) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long long>((n / 4 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(k_plane_to_u8, dim3(grid), dim3(256), 0, s, d_y, n, d_out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_u8_to_plane(const unsigned char* __restrict__ in, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (float)in[i];
}
hipError_t channel8_plane(hipStream_t s, const unsigned char* d_u8, long long n, float* d_out) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_u8_to_plane, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, d_u8, n, d_out);
    return hipGetLastError();
}

// ---- cv::bilateralFilter on a single-channel 8-bit plane (d = -1, BORDER_DEFAULT), as the denoise wrapper
// calls it (src/filter.cpp:371,535).  OpenCV's documented algorithm for CV_8UC1: radius = round(1.5 sigma_space)
// (at least 1), circular window, weight = space[dy,dx] * colour[|v - v0|] from two fp32 tables, fp32 sums in
// row-major window order, result round-half-even of sum / wsum; the border is reflected without repeating the
// edge pixel (BORDER_REFLECT_101).  The tables are built on the host so that host and device forms agree bit
// for bit; products and sums are kept unfused for the same reason.  Planes are fp32 holding integers 0..255.
constexpr int kBfTx = 32, kBfTy = 8;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__global__ __launch_bounds__(kBfTx* kBfTy) void k_bilateral8(const float* __restrict__ src, int H, int W, int radius,
                                                              const float* __restrict__ space_w,
                                                              const float* __restrict__ colour_w,
                                                              float* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int d = 2 * radius + 1, tw = kBfTx + 2 * radius, th = kBfTy + 2 * radius;
    float* tile = reinterpret_cast<float*>(smem_raw);  // [th][tw]
    float* sw = tile + tw * th;                         // [d][d], 0 outside the circle
    float* cw = sw + d * d;                             // [256]
    const int tid = threadIdx.y * kBfTx + threadIdx.x, nthr = kBfTx * kBfTy;
    const int x0 = blockIdx.x * kBfTx - radius, y0 = blockIdx.y * kBfTy - radius;
    for (int i = tid; i < tw * th; i += nthr) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[i] = src[(size_t)reflect101(y0 + ty, H) * W + reflect101(x0 + tx, W)];
    }
    for (int i = tid; i < d * d; i += nthr) sw[i] = space_w[i];
    for (int i = tid; i < 256; i += nthr) cw[i] = colour_w[i];
    __syncthreads();
    const int x = blockIdx.x * kBfTx + threadIdx.x, y = blockIdx.y * kBfTy + threadIdx.y;
    if (x >= W || y >= H) return;
    const float v0 = tile[(threadIdx.y + radius) * tw + threadIdx.x + radius];
    float sum = 0.f, wsum = 0.f;
    for (int dy = 0; dy < d; ++dy) {
        const float* trow = tile + (threadIdx.y + dy) * tw + threadIdx.x;
        const float* srow = sw + dy * d;
        for (int dx = 0; dx < d; ++dx) {
            const float s = srow[dx];
            if (s == 0.f) continue;  // outside the circular window
            const float v = trow[dx];
            const float w = __fmul_rn(s, cw[(int)fabsf(v - v0)]);
            sum = __fadd_rn(sum, __fmul_rn(v, w));
            wsum = __fadd_rn(wsum, w);
        }
    }
    dst[(size_t)y * W + x] = rintf(__fdiv_rn(sum, wsum));
}

int bilateral8_max_radius() { return 64; }

hipError_t bilateral8(hipStream_t s, const float* d_src, int H, int W, int radius, const float* d_space_w,
                      const float* d_colour_w, float* d_dst) {
    if (H <= 0 || W <= 0) return hipSuccess;
    if (radius < 1 || radius > bilateral8_max_radius()) return hipErrorInvalidValue;
    const int d = 2 * radius + 1;
    const size_t shm = ((size_t)(kBfTx + 2 * radius) * (kBfTy + 2 * radius) + (size_t)d * d + 256) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral8),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bilateral8, dim3((unsigned)((W + kBfTx - 1) / kBfTx), (unsigned)((H + kBfTy - 1) / kBfTy)),
                       dim3(kBfTx, kBfTy), shm, s, d_src, H, W, radius, d_space_w, d_colour_w, d_dst);
    return hipGetLastError();
}

// per-block min / max of the first ncols columns of X (M x ld): out[block][2*ncols]
__global__ __launch_bounds__(256) void k_col_range(const float* __restrict__ X, long long M, int ld, int ncols,
                                                   float* __restrict__ out) {
    __shared__ float smn[256], smx[256];
    for (int k = 0; k < ncols; ++k) {
        float mn = 3.4e38f, mx = -3.4e38f;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long long)gridDim.x * 256) {
            const float v = X[(size_t)i * ld + k];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
        smn[threadIdx.x] = mn;
        smx[threadIdx.x] = mx;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + off]);
                smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + off]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[(size_t)blockIdx.x * 2 * ncols + 2 * k] = smn[0];
            out[(size_t)blockIdx.x * 2 * ncols + 2 * k + 1] = smx[0];
        }
        __syncthreads();
    }
}

hipError_t col_range(hipStream_t s, const float* d_X, long long M, int ld, int ncols, float* d_out, int nblocks) {
    hipLaunchKernelGGL(k_col_range, dim3(nblocks), dim3(256), 0, s, d_X, M, ld, ncols, d_out);
    return hipGetLastError();
}

}  // namespace nlek
