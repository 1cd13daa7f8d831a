// Householder tridiagonalisation of a small symmetric matrix (n <= 224) by ONE workgroup, the matrix in registers.
//
// Why it exists, and why it is NOT the default: after the q x q products moved to the device (k_gemm64s), what is left on
// the host between the Sinkhorn loop and the apply are two eigen-computations of order ~p, and VERDICT r1 asked for them
// on the device ("one-workgroup tridiagonal ...").  Measured on the MI355X box at n = 196, K = 50 (tools/host_eig_split.py):
// the host solve is 1.09 ms = reduction 0.44 + QL 0.22 + inverse iteration 0.20 + back-transformation 0.20; this kernel
// does the reduction in 0.66 ms (3.3 us per step: five workgroup barriers, two single-wave sections and ~300 dependent
// fp64 operations per step on ONE compute unit), so with it the solve takes 1.39 ms.  A 4.3 GHz host core wins a
// 200-step latency chain.  The kernel stays as an opt-in (NLE_DEVICE_TRIDIAG=1, nle_eigen_decomposition_top_device) with
// its test; the O(n^2) rest -- QL, inverse iteration for the K kept eigenvectors, their back-transformation -- is the
// host's either way (eigen_sym.cpp: eigen_decomposition_top_reduced).
//
// Same algorithm, storage and scaling as tridiag_reduce in eigen_sym.cpp (the EISPACK tred2 recurrence: rows n-1 .. 1,
// u_i = scaled row i, h_i, p = A u / h, q = p - (u.p / 2h) u, A -= u q^T + q u^T), so the host's back-transformation takes
// its output as is: V (n x n col-major) holds u_i in column i, rows 0..i-1; hs[i] = h_i; d, e the tridiagonal matrix.
// Only the order of the floating-point sums differs from the host form.
//
// Mapping: the matrix is cut into 8 x 8 blocks and only the 406 blocks on and below the diagonal exist, one per thread
// (128 VGPRs; 448 threads = 7 waves, at most two per SIMD).  Per step: the block row of row i publishes it; wave 0 scales it and forms u and h;
// every active thread multiplies its block with u BOTH ways (its rows with u's columns and, off the diagonal, its columns
// with u's rows) and leaves the 8 + 8 partial sums in LDS slots that never collide; 224 threads add the partials of
// their row (p), wave 0 forms q; every active thread applies the rank-2 update to its block.  Five barriers per step.
#include "kernels.h"

namespace nlek {

namespace {
constexpr int kTB = 8;            // block edge
constexpr int kTG = 28;           // blocks per side
constexpr int kTN = kTB * kTG;    // 224
constexpr int kTP = kTN + 1;      // partial-sum row stride (doubles)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
}  // namespace

int tridiag_max_n() { return kTN; }

constexpr int kTT = 448;  // threads: 406 blocks, rounded up to whole waves (7 waves: two per SIMD, 256 VGPRs each)

// Q: n x n column-major, LOWER triangle read (SelfAdjointEigenSolver's convention); diag_add: n values added to the
// diagonal, or null
__global__ __launch_bounds__(kTT) void k_tridiag(int n, const double* __restrict__ Q, const double* __restrict__ diag_add,
                                                 double* __restrict__ V, double* __restrict__ d_out,
                                                 double* __restrict__ e_out, double* __restrict__ hs_out) {
    __shared__ double su[2][kTN];        // u of the step (zeros from index i on), double-buffered by step parity
    __shared__ double sp[kTN];           // p, then q (zeros from index i on)
    __shared__ double spart[kTG * kTP];  // [slot][row] partial products; row r of block row b: slots <= b from the blocks
                                         // of its block row (as rows), slots > b from the blocks of its block column
    __shared__ double sscal[2];          // h, flag (scale == 0)
    const int tid = threadIdx.x;
    // block (ti, tj), tj <= ti, of linear index tid = ti (ti + 1) / 2 + tj
    int ti = (int)((sqrtf(8.0f * (float)tid + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
    while (ti * (ti + 1) / 2 > tid) --ti;
    const int tj = tid - ti * (ti + 1) / 2;
    const bool have = ti < kTG;  // threads 406.. hold no block
    const int R0 = ti * kTB, C0 = tj * kTB;
    const bool diag = ti == tj;
    double a[kTB][kTB];
#pragma unroll
    for (int r = 0; r < kTB; ++r)
#pragma unroll
        for (int c = 0; c < kTB; ++c) {
            const int gr = R0 + r, gc = C0 + c;
            double v = 0.0;
            if (have && gr < n && gc < n) {
                v = (gr >= gc) ? Q[(size_t)gc * n + gr] : Q[(size_t)gr * n + gc];
                if (gr == gc && diag_add != nullptr) v += diag_add[gr];
            }
            a[r][c] = v;
        }
    for (int i = n - 1; i >= 1; --i) {
        const int buf = i & 1;
        double* u = su[buf];
        // 1. row i, columns < i: held by the blocks of block row i / 7
        if (have && ti == i / kTB) {
            const int rr = i - R0;  // uniform
            double row[kTB];
#pragma unroll
            for (int c = 0; c < kTB; ++c) row[c] = 0.0;
#pragma unroll
            for (int r = 0; r < kTB; ++r)
                if (r == rr) {
#pragma unroll
                    for (int c = 0; c < kTB; ++c) row[c] = a[r][c];
                }
#pragma unroll
            for (int c = 0; c < kTB; ++c)
                if (C0 + c < i) u[C0 + c] = row[c];
        }
        __syncthreads();
        // 2. wave 0: scale, h, u (the scaled row with u[i-1] = f - g, zeros from index i on), e[i]
        if (tid < 64) {
            double x[4];
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = tid + 64 * t;
                x[t] = (k < i) ? u[k] : 0.0;
                s += fabs(x[t]);
            }
            const double scale = wave_sum(s);
            if (scale == 0.0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = tid + 64 * t;
                    if (k < i) V[(size_t)i * n + k] = 0.0;
                }
                if (tid == 0) {
                    e_out[i] = u[i - 1];
                    hs_out[i] = 0.0;
                    sscal[0] = 0.0;
                    sscal[1] = 1.0;
                }
            } else {
                double hp = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    x[t] /= scale;
                    hp += x[t] * x[t];
                }
                double h = wave_sum(hp);
                const double f = u[i - 1] / scale;
                double g = sqrt(h);
                if (f > 0) g = -g;
                h -= f * g;
                __builtin_amdgcn_wave_barrier();  // every lane has read u[i-1] before it is overwritten
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = tid + 64 * t;
                    if (k == i - 1) x[t] = f - g;
                    if (k < kTN) u[k] = x[t];
                    if (k < i) V[(size_t)i * n + k] = x[t];
                }
                if (tid == 0) {
                    e_out[i] = scale * g;
                    hs_out[i] = h;
                    sscal[0] = h;
                    sscal[1] = 0.0;
                }
            }
        }
        __syncthreads();
        const bool skip = sscal[1] != 0.0;  // uniform: nothing to eliminate in this row
        const double h = skip ? 1.0 : sscal[0];
        const bool active = have && R0 < i && !skip;  // then C0 <= R0 < i as well
        // 3a. partial products of my block with u, both ways
        if (active) {
            double uc[kTB], ur[kTB];
#pragma unroll
            for (int k = 0; k < kTB; ++k) {
                uc[k] = u[C0 + k];
                ur[k] = u[R0 + k];
            }
#pragma unroll
            for (int r = 0; r < kTB; ++r) {  // rows of the block x u's columns -> slot tj of rows R0..
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < kTB; ++c) s += a[r][c] * uc[c];
                spart[tj * kTP + R0 + r] = s;
            }
            if (!diag) {
#pragma unroll
                for (int c = 0; c < kTB; ++c) {  // columns of the block x u's rows -> slot ti of rows C0..
                    double s = 0.0;
#pragma unroll
                    for (int r = 0; r < kTB; ++r) s += a[r][c] * ur[r];
                    spart[ti * kTP + C0 + c] = s;
                }
            }
        }
        __syncthreads();
        // 3b. p = A u / h
        if (tid < kTN) {
            double s = 0.0;
            if (tid < i) {
                const int nslot = (i + kTB - 1) / kTB;
                for (int t = 0; t < nslot; ++t) s += spart[t * kTP + tid];
                s /= h;
            }
            sp[tid] = s;
        }
        __syncthreads();
        // 4. wave 0: q = p - (u.p / 2h) u
        if (tid < 64) {
            double pk[4], uk[4];
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = tid + 64 * t;
                pk[t] = (k < kTN) ? sp[k] : 0.0;
                uk[t] = (k < kTN) ? u[k] : 0.0;
                s += pk[t] * uk[t];
            }
            const double hh = wave_sum(s) / (h + h);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = tid + 64 * t;
                if (k < kTN) sp[k] = pk[t] - hh * uk[t];
            }
        }
        __syncthreads();
        // 5. A -= u q^T + q u^T on the leading i x i block (u, q vanish from index i on)
        if (active) {
            double uc[kTB], qc[kTB];
#pragma unroll
            for (int k = 0; k < kTB; ++k) {
                uc[k] = u[C0 + k];
                qc[k] = sp[C0 + k];
            }
#pragma unroll
            for (int r = 0; r < kTB; ++r) {
                const double ur = u[R0 + r], qr = sp[R0 + r];
#pragma unroll
                for (int c = 0; c < kTB; ++c) a[r][c] -= ur * qc[c] + qr * uc[c];
            }
        }
        // (the next step writes the other u buffer; spart and sp are rewritten only after its first barriers)
    }
    if (have && diag) {
#pragma unroll
        for (int r = 0; r < kTB; ++r)
            if (R0 + r < n) d_out[R0 + r] = a[r][r];
    }
    if (tid == 0) {
        e_out[0] = 0.0;
        hs_out[0] = 0.0;
    }
}

hipError_t tridiag(hipStream_t s, int n, const double* d_Q, const double* d_diag_add, double* d_V, double* d_d, double* d_e,
                   double* d_hs) {
    if (n < 2 || n > kTN) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tridiag, dim3(1), dim3(kTT), 0, s, n, d_Q, d_diag_add, d_V, d_d, d_e, d_hs);
    return hipGetLastError();
}

}  // namespace nlek
