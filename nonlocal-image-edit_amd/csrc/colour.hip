// 8-bit BGR <-> Lab on the device: the colour wrapper either side of the hot path
// (reference src/filter.cpp:460-469 getLuminanceChannel, :422-426 and :434-440 in NLEFilter::enhance,
// where it is cv::cvtColor on 8-bit images).  BGR -> Lab is OpenCV's fixed-point table algorithm (exact integers, the
// same as the host restatement in host/filter.cpp); Lab -> BGR the documented float formula in fp64 (agrees with the host
// restatement except for isolated round-to-nearest ties; OpenCV's own 8-bit inverse is not pinned).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace nlek {

namespace {
__device__ __forceinline__ unsigned char sat8(double v) {
    return (unsigned char)fmin(255.0, fmax(0.0, rint(v)));  // round half to even, saturate (cv convertTo CV_8U)
}
__device__ __forceinline__ double lin2srgb(double v) {
    return v <= 0.0031308 ? 12.92 * v : 1.055 * pow(fmax(v, 0.0), 1.0 / 2.4) - 0.055;
}
}  // namespace

// OpenCV's fixed-point 8-bit path (imgproc RGB2Lab_b, restated in include/nle.h at nle_lab8_tables): tables in LDS, integer
// arithmetic throughout -- bit-identical to the host restatement and to the oracle.  lut = gamma[256] | cbrt[3072] (u16)
// | coeffs[9] (int) as abi_ctx.hip uploads them; d_L (optional) = L channel as fp32
__global__ __launch_bounds__(256) void k_bgr2lab8(const unsigned char* __restrict__ bgr, long long n,
                                                  const double* __restrict__ lut, unsigned char* __restrict__ lab,
                                                  float* __restrict__ Lf) {
    constexpr int NCB = 256 * 3 / 2 * 8;
    __shared__ unsigned short sg[256], sc[NCB];
    __shared__ int sk[9];
    const unsigned short* t16 = reinterpret_cast<const unsigned short*>(lut);
    for (int i = threadIdx.x; i < 256; i += 256) sg[i] = t16[i];
    for (int i = threadIdx.x; i < NCB; i += 256) sc[i] = t16[256 + i];
    if (threadIdx.x < 9) sk[threadIdx.x] = reinterpret_cast<const int*>(t16 + 256 + NCB)[threadIdx.x];
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
// (the denoise wrapper also replaces a and b by filtered planes, clamped and rounded the same way, :391-399)
__global__ __launch_bounds__(256) void k_lab2bgr8(const unsigned char* __restrict__ lab, const float* __restrict__ Lf,
                                                  const float* __restrict__ af, const float* __restrict__ bf,
                                                  long long n, unsigned char* __restrict__ bgr) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double L8 = Lf != nullptr ? (double)sat8((double)Lf[i]) : (double)lab[3 * i + 0];
        const double a8 = af != nullptr ? (double)sat8((double)af[i]) : (double)lab[3 * i + 1];
        const double b8 = bf != nullptr ? (double)sat8((double)bf[i]) : (double)lab[3 * i + 2];
        const double L = L8 * 100.0 / 255.0, a = a8 - 128.0, b = b8 - 128.0;
        double fy = (L + 16.0) / 116.0, y;
        if (L > 7.9996248) {
            y = fy * fy * fy;
        } else {
            y = L / 903.3;
            fy = 7.787 * y + 16.0 / 116.0;
        }
        const double fx = a / 500.0 + fy, fz = fy - b / 200.0;
        const double x = (fx > 0.206893 ? fx * fx * fx : (fx - 16.0 / 116.0) / 7.787) * 0.950456;
        const double z = (fz > 0.206893 ? fz * fz * fz : (fz - 16.0 / 116.0) / 7.787) * 1.088754;
        const double r = 3.240479 * x - 1.53715 * y - 0.498535 * z;
        const double g = -0.969256 * x + 1.875991 * y + 0.041556 * z;
        const double bl = 0.055648 * x - 0.204043 * y + 1.057311 * z;
        bgr[3 * i + 0] = sat8(lin2srgb(fmin(1.0, fmax(0.0, bl))) * 255.0);
        bgr[3 * i + 1] = sat8(lin2srgb(fmin(1.0, fmax(0.0, g))) * 255.0);
        bgr[3 * i + 2] = sat8(lin2srgb(fmin(1.0, fmax(0.0, r))) * 255.0);
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
                    long long n, unsigned char* d_bgr) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_lab2bgr8, dim3(grid), dim3(256), 0, s, d_lab, d_L, d_a, d_b, n, d_bgr);
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
