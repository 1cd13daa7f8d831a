// gfx950 (MI355X / CDNA4) kernels of the nonlocal-filter hot path.
//
// Data layout: every N-sized matrix is fp32, one ROW PER PIXEL (row-major, leading
// dimension ld % 4 == 0, pad columns zero), pixels in natural image order.  Every
// reduction over pixels accumulates in fp64 (two-stage, fixed order => deterministic);
// the tall-skinny products run on the fp32-input MFMA (v_mfma_f32_32x32x2_f32, an exact
// fp32 fma chain) with fp32 accumulation over at most a few thousand rows.
//
// Reference arithmetic each kernel replaces (reference tree, src/filter.cpp):
//   k_affinity      :104-112,139-145   Kab(i,j) = exp(negativeWeightedDistance)
//   k_tsgemm<FUSED> :275               Kab^T * eigvecs * invEigVals   (affinity fused)
//   k_tsgemm        :327               tmp * invRootWa * Vq * invRootSq (as diag(c) Phi C)
//   k_rowpass       :239,243 + :42-54  phi*(D*(phi^T*r)) and inplaceReciprocal, one pass
//                   :456               m_eigvecs^T * c
//   k_gram          :296               Wab * Wab^T (its N-sized part, Phi^T C^2 Phi)
//   k_apply_expand  :456               m_eigvecs * (diag * ...)
#include "kernels.h"

namespace nlek {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double recip_or_zero(double s, double eps) {
    // inplaceReciprocal, src/filter.cpp:42-54
    return (fabs(s) >= eps) ? 1.0 / s : 0.0;
}

// ------------------------------------------------------------------ gather samples
__global__ void k_gather_samples(const float* __restrict__ lum, GridSpec gs, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= gs.p()) return;
    const int ri = k / gs.nSelCols, ci = k % gs.nSelCols;
    const int r = gs.rowOff + ri * gs.rowStep, c = gs.colOff + ci * gs.colStep;
    out[k] = lum[(size_t)r * gs.W + c];
}

// the same for a plane of which only rows [row0, row1) exist (lum is the virtual base of the full image): samples of
// other rows come out as 0, so that a sum over the ranks' results (all-reduce) gives every rank all p values; fp64
__global__ void k_gather_samples_slab(const float* __restrict__ lum, GridSpec gs, int row0, int row1,
                                      double* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= gs.p()) return;
    const int ri = k / gs.nSelCols, ci = k % gs.nSelCols;
    const int r = gs.rowOff + ri * gs.rowStep, c = gs.colOff + ci * gs.colStep;
    out[k] = (r >= row0 && r < row1) ? (double)lum[(size_t)r * gs.W + c] : 0.0;
}

hipError_t gather_samples_slab(hipStream_t s, const float* d_lum, GridSpec gs, int row0, int row1, double* d_out) {
    const int p = gs.p();
    hipLaunchKernelGGL(k_gather_samples_slab, dim3((p + 255) / 256), dim3(256), 0, s, d_lum, gs, row0, row1, d_out);
    return hipGetLastError();
}

hipError_t gather_samples(hipStream_t s, const float* d_lum, GridSpec gs, float* d_out) {
    const int p = gs.p();
    hipLaunchKernelGGL(k_gather_samples, dim3((p + 255) / 256), dim3(256), 0, s, d_lum, gs, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ affinity (K_AB)
// Write-bound: reads 4 B, writes 4*ld B per pixel.  Each thread produces one float4 of
// one pixel's row; a block covers 128 consecutive pixels so stores are fully coalesced
// (the 128 rows are one contiguous 128*ld*4-byte span).  Samples live in LDS.
constexpr int kAffPix = 128;

__global__ __launch_bounds__(256) void k_affinity(const float* __restrict__ lum, GridSpec gs,
                                                  const Sample4* __restrict__ samples, int p,
                                                  int ld, float nsw, float npw, unsigned pix0,
                                                  long long M, float* __restrict__ kab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Sample4* ss = reinterpret_cast<Sample4*>(smem_raw);
    for (int k = threadIdx.x; k < ld; k += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < p) v = samples[k];
        ss[k] = v;
    }
    __syncthreads();
    const unsigned nq = (unsigned)ld >> 2;
    const unsigned per_group = kAffPix * nq;
    const long long ngroups = (M + kAffPix - 1) / kAffPix;
    for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long long i0 = grp * kAffPix;
        float* __restrict__ out = kab + i0 * ld;
        const unsigned nvalid = (unsigned)min((long long)kAffPix, M - i0) * nq;
        for (unsigned f = threadIdx.x; f < per_group; f += 256) {
            if (f >= nvalid) break;
            const unsigned il = f / nq, q = f - il * nq;
            const unsigned gi = pix0 + (unsigned)i0 + il;
            const unsigned row = gi / (unsigned)gs.W;
            const float pr = (float)row, pc = (float)(gi - row * (unsigned)gs.W);
            const float px = lum[gi];
            float4 o;
            const unsigned s0 = 4 * q;
            o.x = (s0 + 0 < (unsigned)p) ? affinity_value(pr, pc, px, ss[s0 + 0], nsw, npw) : 0.f;
            o.y = (s0 + 1 < (unsigned)p) ? affinity_value(pr, pc, px, ss[s0 + 1], nsw, npw) : 0.f;
            o.z = (s0 + 2 < (unsigned)p) ? affinity_value(pr, pc, px, ss[s0 + 2], nsw, npw) : 0.f;
            o.w = (s0 + 3 < (unsigned)p) ? affinity_value(pr, pc, px, ss[s0 + 3], nsw, npw) : 0.f;
            *reinterpret_cast<float4*>(out + (size_t)f * 4) = o;
        }
    }
}

hipError_t affinity(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples,
                    int p, int ld, float nsw, float npw, long long pix0, long long M, float* d_kab) {
    if (M <= 0) return hipSuccess;
    const long long ngroups = (M + kAffPix - 1) / kAffPix;
    const int grid = (int)min(ngroups, (long long)8192);
    hipLaunchKernelGGL(k_affinity, dim3(grid), dim3(256), (size_t)ld * sizeof(Sample4), s, d_lum, gs,
                       d_samples, p, ld, nsw, npw, (unsigned)pix0, M, d_kab);
    return hipGetLastError();
}

// ------------------------------------------------------------------ tall-skinny GEMM
// C (M x ldc) = rowscale o (A (M x kd) * B (kd x ldc)) on v_mfma_f32_32x32x2_f32.
// Block = 4 waves = 128 rows; each wave owns 32 rows x NT column tiles of 32 (NT*16
// accumulator registers).  FUSED: the A fragment (lane l: row l&31, k = l>>5) is the
// affinity exp(...) computed in registers -- K_AB never exists in memory.  Otherwise the
// A chunk is staged through LDS (128 x 32, padded rows).  B chunk (32 x NT*32) in LDS.
struct TsArgs {
    const float* A;
    int lda;
    const float* lum;
    GridSpec gs;
    const Sample4* samples;
    float nsw, npw;
    unsigned pix0;
    const float* B;
    int ldb;
    int kd;
    float* C;
    int ldc;
    long long M;
    const double* u;
    double eps;
    const float* cvec;
};

template <int NT, bool FUSED>
__global__ __launch_bounds__(256) void k_tsgemm(TsArgs a) {
    constexpr int KB = 32;
    constexpr int PW = NT * 32;
    __shared__ __attribute__((aligned(16))) float sB[KB][PW];
    __shared__ float sA[FUSED ? 1 : 128][FUSED ? 1 : KB + 1];
    __shared__ Sample4 sS[KB];
    __shared__ float sScale[128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 128;
    const int col0 = blockIdx.y * PW;

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

    float pr = 0.f, pc = 0.f, px = 0.f;
    if constexpr (FUSED) {
        long long myrow = m0 + wave * 32 + l31;
        if (myrow >= a.M) myrow = a.M - 1;
        const unsigned gi = a.pix0 + (unsigned)myrow;
        const unsigned row = gi / (unsigned)a.gs.W;
        pr = (float)row;
        pc = (float)(gi - row * (unsigned)a.gs.W);
        px = a.lum[gi];
    }
    double dpart[4] = {0.0, 0.0, 0.0, 0.0};
    const bool scale = (!FUSED) && (a.u != nullptr);

    for (int k0 = 0; k0 < a.kd; k0 += KB) {
        __syncthreads();
        for (int idx = tid; idx < KB * PW / 4; idx += 256) {
            const int kk = idx / (PW / 4), c4 = idx % (PW / 4);
            const int k = k0 + kk, col = col0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < a.kd && col < a.ldb) v = *reinterpret_cast<const float4*>(a.B + (size_t)k * a.ldb + col);
            *reinterpret_cast<float4*>(&sB[kk][4 * c4]) = v;
        }
        if constexpr (FUSED) {
            if (tid < KB) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + tid < a.kd) v = a.samples[k0 + tid];
                sS[tid] = v;
            }
        } else {
            const int q = tid & 7;
            const int k = k0 + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (tid >> 3) + 32 * j;
                const long long grow = m0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (grow < a.M && k < a.lda) v = *reinterpret_cast<const float4*>(a.A + (size_t)grow * a.lda + k);
                sA[row][4 * q + 0] = v.x;
                sA[row][4 * q + 1] = v.y;
                sA[row][4 * q + 2] = v.z;
                sA[row][4 * q + 3] = v.w;
                if (scale && k < a.lda) {
                    const double* uu = a.u + k;
                    dpart[j] += (double)v.x * uu[0] + (double)v.y * uu[1] + (double)v.z * uu[2] + (double)v.w * uu[3];
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < KB; kk += 2) {
            const int kl = kk + half;
            float av;
            if constexpr (FUSED) {
                av = affinity_value(pr, pc, px, sS[kl], a.nsw, a.npw);
            } else {
                av = sA[wave * 32 + l31][kl];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float bv = sB[kl][n * 32 + l31];
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
            }
        }
    }

    if constexpr (FUSED) {
        if (a.cvec != nullptr && tid < 128) sScale[tid] = (m0 + tid < a.M) ? a.cvec[m0 + tid] : 0.f;
    }
    const bool rowscale = scale || (FUSED && a.cvec != nullptr);
    if (scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = dpart[j];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if ((tid & 7) == 0) sScale[(tid >> 3) + 32 * j] = (float)recip_or_zero(s, a.eps);
        }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = col0 + n * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            const long long grow = m0 + rl;
            if (grow < a.M && col < a.ldc) {
                float v = acc[n][e];
                if (rowscale) v *= sScale[rl];
                a.C[(size_t)grow * a.ldc + col] = v;
            }
        }
    }
}

template <bool FUSED>
static hipError_t launch_tsgemm(hipStream_t s, const TsArgs& a) {
    const int ntiles = (a.ldc + 31) / 32;
    const int panels = (ntiles + 7) / 8;
    const int nt = (ntiles + panels - 1) / panels;
    const dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)panels), block(256);
    switch (nt) {
#define NLE_TS_CASE(NTV)                                                              \
    case NTV:                                                                         \
        hipLaunchKernelGGL((k_tsgemm<NTV, FUSED>), grid, block, 0, s, a);             \
        break;
        NLE_TS_CASE(1)
        NLE_TS_CASE(2)
        NLE_TS_CASE(3)
        NLE_TS_CASE(4)
        NLE_TS_CASE(5)
        NLE_TS_CASE(6)
        NLE_TS_CASE(7)
        NLE_TS_CASE(8)
#undef NLE_TS_CASE
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t ts_gemm(hipStream_t s, bool fused, const float* d_A, int lda, const float* d_lum,
                   GridSpec gs, const Sample4* d_samples, float nsw, float npw, long long pix0,
                   const float* d_B, int ldb, int kd, float* d_C, int ldc, long long M,
                   const double* d_u, double eps, const float* d_c) {
    if (M <= 0) return hipSuccess;
    TsArgs a;
    a.A = d_A;
    a.lda = lda;
    a.lum = d_lum;
    a.gs = gs;
    a.samples = d_samples;
    a.nsw = nsw;
    a.npw = npw;
    a.pix0 = (unsigned)pix0;
    a.B = d_B;
    a.ldb = ldb;
    a.kd = kd;
    a.C = d_C;
    a.ldc = ldc;
    a.M = M;
    a.u = d_u;
    a.eps = eps;
    a.cvec = d_c;
    return fused ? launch_tsgemm<true>(s, a) : launch_tsgemm<false>(s, a);
}

// ------------------------------------------------------------------ row passes
// G lanes share one row (lane lg of the group covers float4 quads lg, lg+G, ...), so a
// wave instruction reads G*16 contiguous bytes of each of 64/G rows; QPL quads per lane.
static bool pick_gq(int ld, int* G, int* Q) {
    const int nq = ld / 4;
    static const int gs[5] = {4, 8, 16, 32, 64};
    for (int i = 0; i < 5; ++i) {
        if (gs[i] * 8 >= nq) {
            int q = (nq + gs[i] - 1) / gs[i];
            if (q == 3) q = 4;
            if (q == 5) q = 6;
            if (q < 1) q = 1;
            *G = gs[i];
            *Q = q;
            return true;
        }
    }
    return false;
}

#define NLE_DISPATCH_Q(GV, Q, ...)                                      \
    switch (Q) {                                                        \
        case 1: { constexpr int G_ = GV, Q_ = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int G_ = GV, Q_ = 2; __VA_ARGS__; } break;  \
        case 4: { constexpr int G_ = GV, Q_ = 4; __VA_ARGS__; } break;  \
        case 6: { constexpr int G_ = GV, Q_ = 6; __VA_ARGS__; } break;  \
        case 7: { constexpr int G_ = GV, Q_ = 7; __VA_ARGS__; } break;  \
        case 8: { constexpr int G_ = GV, Q_ = 8; __VA_ARGS__; } break;  \
        default: return hipErrorInvalidValue;                           \
    }
#define NLE_DISPATCH_GQ(G, Q, ...)                                      \
    switch (G) {                                                        \
        case 4: NLE_DISPATCH_Q(4, Q, __VA_ARGS__) break;                \
        case 8: NLE_DISPATCH_Q(8, Q, __VA_ARGS__) break;                \
        case 16: NLE_DISPATCH_Q(16, Q, __VA_ARGS__) break;              \
        case 32: NLE_DISPATCH_Q(32, Q, __VA_ARGS__) break;              \
        case 64: NLE_DISPATCH_Q(64, Q, __VA_ARGS__) break;              \
        default: return hipErrorInvalidValue;                           \
    }

template <int G, int QPL>
__global__ __launch_bounds__(256) void k_rowpass(int mode, const float* __restrict__ X, long long M,
                                                 int ld, const double* __restrict__ t_in,
                                                 const double* __restrict__ lam,
                                                 const float* __restrict__ xvec, double eps,
                                                 double* __restrict__ partial) {
    constexpr int RW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sred = reinterpret_cast<double*>(smem_raw);  // [4][ld]
    double* su = sred + 4 * ld;                           // [G*QPL*4] (u, zero padded)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, lg = lane % G;
    const int nq = ld >> 2;

    for (int j = threadIdx.x; j < G * QPL * 4; j += 256)
        su[j] = (mode == ROWPASS_RECIP && j < ld) ? lam[j] * t_in[j] : 0.0;
    __syncthreads();
    double acc[QPL][4];
#pragma unroll
    for (int k = 0; k < QPL; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = 0.0;

    const long long step = (long long)gridDim.x * (4 * RW);
    for (long long base = (long long)blockIdx.x * (4 * RW); base < M; base += step) {
        const long long row = base + wave * RW + g;
        const bool valid = row < M;
        float4 v[QPL];
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int q = lg + G * k;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && q < nq) v[k] = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 4 * q);
        }
        double y;
        if (mode == ROWPASS_RECIP) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
                const double* uu = su + 4 * (lg + G * k);
                s += (double)v[k].x * uu[0] + (double)v[k].y * uu[1] + (double)v[k].z * uu[2] +
                     (double)v[k].w * uu[3];
            }
#pragma unroll
            for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off);
            y = recip_or_zero(s, eps);
        } else if (mode == ROWPASS_COLSUM) {
            y = 1.0;
        } else {
            y = valid ? (double)xvec[row] : 0.0;
        }
        if (!valid) y = 0.0;
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            acc[k][0] += (double)v[k].x * y;
            acc[k][1] += (double)v[k].y * y;
            acc[k][2] += (double)v[k].z * y;
            acc[k][3] += (double)v[k].w * y;
        }
    }

#pragma unroll
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < QPL; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[k][c] += __shfl_xor(acc[k][c], off);
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int q = lg + G * k;
            if (q < nq)
#pragma unroll
                for (int c = 0; c < 4; ++c) sred[wave * ld + 4 * q + c] = acc[k][c];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < ld; j += 256)
        partial[(size_t)blockIdx.x * ld + j] = (sred[j] + sred[ld + j]) + (sred[2 * ld + j] + sred[3 * ld + j]);
}

hipError_t rowpass(hipStream_t s, int mode, const float* d_X, long long M, int ld,
                   const double* d_t_in, const double* d_lam, const float* d_xvec, double eps,
                   double* d_partial, int* nblocks) {
    int G, Q;
    if (!pick_gq(ld, &G, &Q)) return hipErrorInvalidValue;
    const int rows_per_block = 4 * (64 / G);
    long long nb = (M + rows_per_block - 1) / rows_per_block;
    if (nb > kRowpassMaxBlocks) nb = kRowpassMaxBlocks;
    if (nb < 1) nb = 1;
    *nblocks = (int)nb;
    const size_t shm = ((size_t)4 * ld + (size_t)G * Q * 4) * sizeof(double);
    NLE_DISPATCH_GQ(G, Q,
                    hipLaunchKernelGGL((k_rowpass<G_, Q_>), dim3((unsigned)nb), dim3(256), shm, s, mode,
                                       d_X, M, ld, d_t_in, d_lam, d_xvec, eps, d_partial))
    return hipGetLastError();
}

// partial [nb][ld] -> t_out[nslices][ld]; block (x, y) = 32 columns x the y-th contiguous slice of
// the rows, 8 threads per column each summing every 8th row of the slice (fixed order)
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partial, int nb,
                                                         int ld, double* __restrict__ t_out) {
    __shared__ double sm[8][32];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    const int per = (nb + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nb, b0 + per);
    double s = 0.0;
    if (col < ld)
        for (int b = b0 + sl; b < b1; b += 8) s += partial[(size_t)b * ld + col];
    sm[sl][c] = s;
    __syncthreads();
    if (sl == 0 && col < ld) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][c];
        t_out[(size_t)blockIdx.y * ld + col] = t;
    }
}

hipError_t reduce_partials(hipStream_t s, const double* d_partial, int nblocks, int ld,
                           double* d_t_out, int nslices) {
    hipLaunchKernelGGL(k_reduce_partials, dim3((ld + 31) / 32, nslices), dim3(256), 0, s, d_partial, nblocks, ld,
                       d_t_out);
    return hipGetLastError();
}

__global__ void k_scale_vec(const double* __restrict__ lam, const double* __restrict__ t, int n,
                            double* __restrict__ u) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) u[j] = lam[j] * t[j];
}

hipError_t scale_vec(hipStream_t s, const double* d_lam, const double* d_t, int n, double* d_u) {
    hipLaunchKernelGGL(k_scale_vec, dim3((n + 255) / 256), dim3(256), 0, s, d_lam, d_t, n, d_u);
    return hipGetLastError();
}

template <int G, int QPL>
__global__ __launch_bounds__(256) void k_row_scalings(const float* __restrict__ X, long long M, int ld,
                                                      const double* __restrict__ uvec, double eps,
                                                      double* __restrict__ out) {
    constexpr int RW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, lg = lane % G;
    const int nq = ld >> 2;
    const long long step = (long long)gridDim.x * (4 * RW);
    for (long long base = (long long)blockIdx.x * (4 * RW); base < M; base += step) {
        const long long row = base + wave * RW + g;
        const bool valid = row < M;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int q = lg + G * k;
            if (valid && q < nq) {
                const float4 v = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 4 * q);
                const double* uu = uvec + 4 * q;
                s += (double)v.x * uu[0] + (double)v.y * uu[1] + (double)v.z * uu[2] + (double)v.w * uu[3];
            }
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off);
        if (valid && lg == 0) out[row] = recip_or_zero(s, eps);
    }
}

hipError_t row_scalings(hipStream_t s, const float* d_X, long long M, int ld, const double* d_u,
                        double eps, double* d_out) {
    if (M <= 0) return hipSuccess;
    int G, Q;
    if (!pick_gq(ld, &G, &Q)) return hipErrorInvalidValue;
    const int rows_per_block = 4 * (64 / G);
    long long nb = (M + rows_per_block - 1) / rows_per_block;
    if (nb > 2048) nb = 2048;
    NLE_DISPATCH_GQ(G, Q,
                    hipLaunchKernelGGL((k_row_scalings<G_, Q_>), dim3((unsigned)nb), dim3(256), 0, s, d_X, M,
                                       ld, d_u, eps, d_out))
    return hipGetLastError();
}

// ------------------------------------------------------------------ Gram
// G = sum_i c_i^2 x_i x_i^T.  grid.x = row chunks (fp32 MFMA accumulation stays inside
// one chunk of <= 8192 rows; chunk partials are summed in fp64 by k_gram_reduce),
// grid.y = groups of 28 upper-triangular 32x32 tiles (7 per wave).  Rows are staged
// 32 at a time into LDS already scaled by c_i (8 lanes per row compute x_i . u in fp64).
int gram_num_tiles(int ld) {
    const int nt = (ld + 31) / 32;
    return nt * (nt + 1) / 2;
}
int gram_chunk_rows(long long M) {
    long long fl = (M + 1023) / 1024;
    fl = ((fl + 31) / 32) * 32;
    if (fl < 256) fl = 256;
    if (fl > 8192) fl = 8192;
    return (int)fl;
}
static long long gram_num_chunks(long long M) {
    const int fl = gram_chunk_rows(M);
    return (M + fl - 1) / fl;
}
size_t gram_partial_elems(long long M, int ld) {
    return (size_t)gram_num_chunks(M) * gram_num_tiles(ld) * 1024;
}

__device__ __forceinline__ void tile_coords(int t, int nt, int* ti, int* tj) {
    int i = 0;
    while (t >= nt - i) {
        t -= nt - i;
        ++i;
    }
    *ti = i;
    *tj = i + t;
}

__global__ __launch_bounds__(256) void k_gram(const float* __restrict__ X, long long M, int ld,
                                              const double* __restrict__ uvec, double eps,
                                              int chunk_rows, int ntiles, double* __restrict__ partial) {
    constexpr int RB = kGramRowsPerStage, TPW = kGramTilesPerWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* su = reinterpret_cast<double*>(smem_raw);                       // [ld]
    float* sZ = reinterpret_cast<float*>(smem_raw + (size_t)ld * sizeof(double));  // [RB][ld]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int nq = ld >> 2, nt = (ld + 31) / 32;

    int ca[TPW], cb[TPW];
    bool tv[TPW];
    const int tbase = (blockIdx.y * 4 + wave) * TPW;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = tbase + j;
        tv[j] = t < ntiles;
        int ti = 0, tj = 0;
        if (tv[j]) tile_coords(t, nt, &ti, &tj);
        ca[j] = ti * 32 + l31;
        cb[j] = tj * 32 + l31;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    for (int j = tid; j < ld; j += 256) su[j] = uvec ? uvec[j] : 0.0;

    const long long c0 = (long long)blockIdx.x * chunk_rows;
    const long long c1 = min(M, c0 + (long long)chunk_rows);
    const int g = lane >> 3, lg = lane & 7;
    const int rl = wave * 8 + g;
    for (long long rb = c0; rb < c1; rb += RB) {
        __syncthreads();
        const long long row = rb + rl;
        const bool valid = row < c1;
        float* zrow = sZ + (size_t)rl * ld;
        double s = 0.0;
        for (int q0 = lg; q0 < nq; q0 += 32) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + 8 * k;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid && q < nq) v[k] = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 4 * q);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + 8 * k;
                if (q < nq) {
                    *reinterpret_cast<float4*>(zrow + 4 * q) = v[k];
                    const double* uu = su + 4 * q;
                    s += (double)v[k].x * uu[0] + (double)v[k].y * uu[1] + (double)v[k].z * uu[2] +
                         (double)v[k].w * uu[3];
                }
            }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        const float cf = valid ? (uvec ? (float)recip_or_zero(s, eps) : 1.f) : 0.f;
        for (int q = lg; q < nq; q += 8) {
            float4* pz = reinterpret_cast<float4*>(zrow + 4 * q);
            float4 v = *pz;
            v.x *= cf;
            v.y *= cf;
            v.z *= cf;
            v.w *= cf;
            *pz = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < RB; kk += 2) {
            const float* zr = sZ + (size_t)(kk + half) * ld;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (tv[j]) {
                    const float av = (ca[j] < ld) ? zr[ca[j]] : 0.f;
                    const float bv = (cb[j] < ld) ? zr[cb[j]] : 0.f;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (tv[j]) {
            double* out = partial + ((size_t)blockIdx.x * ntiles + (tbase + j)) * 1024;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * half;
                out[r * 32 + l31] = (double)acc[j][e];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ partial, int nchunks,
                                                     int ntiles, double* __restrict__ tiles) {
    const size_t e = (size_t)blockIdx.x * 1024 + blockIdx.y * 256 + threadIdx.x;
    const size_t stride = (size_t)ntiles * 1024;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = 0;
    for (; c + 3 < nchunks; c += 4) {
        s0 += partial[(size_t)c * stride + e];
        s1 += partial[(size_t)(c + 1) * stride + e];
        s2 += partial[(size_t)(c + 2) * stride + e];
        s3 += partial[(size_t)(c + 3) * stride + e];
    }
    for (; c < nchunks; ++c) s0 += partial[(size_t)c * stride + e];
    tiles[e] = (s0 + s1) + (s2 + s3);
}

hipError_t gram(hipStream_t s, const float* d_X, long long M, int ld, const double* d_u, double eps,
                double* d_partial, double* d_tiles) {
    const int ntiles = gram_num_tiles(ld);
    const int fl = gram_chunk_rows(M);
    const long long nchunks = gram_num_chunks(M);
    const int groups = (ntiles + 4 * kGramTilesPerWave - 1) / (4 * kGramTilesPerWave);
    const size_t shm = (size_t)ld * sizeof(double) + (size_t)kGramRowsPerStage * ld * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gram, dim3((unsigned)nchunks, (unsigned)groups), dim3(256), shm, s, d_X, M, ld, d_u,
                       eps, fl, ntiles, d_partial);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return gram_reduce(s, d_partial, (int)nchunks, ntiles, d_tiles);
}

hipError_t gram_reduce(hipStream_t s, const double* d_partial, int nchunks, int ntiles, double* d_tiles) {
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)ntiles, 4), dim3(256), 0, s, d_partial, nchunks, ntiles,
                       d_tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------ apply: expand
template <int G, int QPL>
__global__ __launch_bounds__(256) void k_apply_expand(const float* __restrict__ V, long long M, int ld,
                                                      const double* __restrict__ gm, int L,
                                                      float* __restrict__ Y, long long ystride) {
    constexpr int RW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sg = reinterpret_cast<double*>(smem_raw);  // [L][ld]
    for (int j = threadIdx.x; j < L * ld; j += 256) sg[j] = gm[j];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, lg = lane % G;
    const int nq = ld >> 2;
    const long long step = (long long)gridDim.x * (4 * RW);
    for (long long base = (long long)blockIdx.x * (4 * RW); base < M; base += step) {
        const long long row = base + wave * RW + g;
        const bool valid = row < M;
        float4 v[QPL];
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            const int q = lg + G * k;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && q < nq) v[k] = *reinterpret_cast<const float4*>(V + (size_t)row * ld + 4 * q);
        }
        for (int l = 0; l < L; ++l) {
            const double* gl = sg + (size_t)l * ld;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < QPL; ++k) {
                const int q = lg + G * k;
                if (q < nq) {
                    const double* gg = gl + 4 * q;
                    s += (double)v[k].x * gg[0] + (double)v[k].y * gg[1] + (double)v[k].z * gg[2] +
                         (double)v[k].w * gg[3];
                }
            }
#pragma unroll
            for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off);
            if (valid && lg == 0) Y[(size_t)l * ystride + row] = (float)s;
        }
    }
}

hipError_t apply_expand(hipStream_t s, const float* d_V, long long M, int ld, const double* d_g, int L,
                        float* d_Y, long long ystride) {
    if (M <= 0) return hipSuccess;
    int G, Q;
    if (!pick_gq(ld, &G, &Q)) return hipErrorInvalidValue;
    const int rows_per_block = 4 * (64 / G);
    long long nb = (M + rows_per_block - 1) / rows_per_block;
    if (nb > 4096) nb = 4096;
    const size_t shm = (size_t)L * ld * sizeof(double);
    NLE_DISPATCH_GQ(G, Q,
                    hipLaunchKernelGGL((k_apply_expand<G_, Q_>), dim3((unsigned)nb), dim3(256), shm, s, d_V,
                                       M, ld, d_g, L, d_Y, ystride))
    return hipGetLastError();
}

// ------------------------------------------------------------------ scatter rows
__global__ void k_scatter_rows(const float* __restrict__ src, const long long* __restrict__ idx, int n,
                               int ld, float* __restrict__ X, long long M) {
    const int k = blockIdx.x;
    if (k >= n) return;
    const long long row = idx[k];
    if (row < 0 || row >= M) return;
    for (int j = threadIdx.x; j < ld; j += blockDim.x) X[(size_t)row * ld + j] = src[(size_t)k * ld + j];
}

hipError_t scatter_rows(hipStream_t s, const float* d_src, const long long* d_idx, int n, int ld,
                        float* d_X, long long M) {
    if (n <= 0 || M <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_rows, dim3(n), dim3(128), 0, s, d_src, d_idx, n, ld, d_X, M);
    return hipGetLastError();
}

}  // namespace nlek
