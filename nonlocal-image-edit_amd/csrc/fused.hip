// Phi-free ("sample space") kernels of the fused train path for gfx950.
//
// Identity used (all exact): with B = V_A diag(1/lambda) (p x r), the Nystrom row of a
// non-sample pixel i is phi_i = B^T k_i, k_i[s] = exp(negDist(pixel i, sample s))
// (reference src/filter.cpp:139-145,275).  Hence
//     phi_i . u          = k_i . (B u)                         (Sinkhorn row product, :239,243)
//     Phi^T y            = B^T (sum_i k_i y_i)                 (Sinkhorn column sums)
//     sum c_i^2 phi phi^T = B^T (sum_i c_i^2 k_i k_i^T) B       (Gram of :296)
//     c_i phi_i C        = c_i k_i^T (B C)                     (projection :327)
// so every N-sized pass can regenerate its affinity row in registers (12 B/pixel of HBM
// traffic) instead of streaming an N x r matrix: the passes become ALU-bound at ~1/4 of
// the time it takes to read Phi once at HBM speed, and Phi is never allocated.
// Sample pixels (whose rows are the exact V_A rows, :275 top block) are skipped by the
// N-sized kernels and handled in fp64 by k_sink_update / the host.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace nlek {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double recip_or_zero_d(double s, double eps) {
    return (fabs(s) >= eps) ? 1.0 / s : 0.0;  // inplaceReciprocal, src/filter.cpp:42-54
}

// ------------------------------------------------------------------ wave transpose-reduce
// v[N] per lane (N % 64 == 0).  On return out[j] of lane l = sum over the 64 lanes of
// v[64 j + l].  log2(64) levels; level with lane bit b halves the array: the lane keeps the
// half selected by its bit b and adds the partner lane's (l ^ (1 << b)) copy of that half.
template <int MASK>
__device__ __forceinline__ void fold_level(float& a, float& b, int lane) {
    // a <- (lane & MASK ? b : a) summed over the lane pair {l, l ^ MASK}
    if constexpr (MASK == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (MASK == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        const bool hi = (lane & MASK) != 0;
        const float keep = hi ? b : a, give = hi ? a : b;
        a = keep + __shfl_xor(give, MASK);
    }
}

template <int N, int LEN, int W>
__device__ __forceinline__ void fold_all(float (&v)[N], int lane) {
    // entries [0, LEN) are live, laid out [j][t] with t < W
    if constexpr (W > 1) {
        constexpr int H = W / 2;
#pragma unroll
        for (int j = 0; j < LEN / W; ++j)
#pragma unroll
            for (int t = 0; t < H; ++t) {
                float a = v[j * W + t], b = v[j * W + t + H];
                fold_level<H>(a, b, lane);
                v[j * H + t] = a;  // compacts to [j][t] at width H; j*H+t is never read again
            }
        fold_all<N, LEN / 2, H>(v, lane);
    }
}

// ------------------------------------------------------------------ Sinkhorn half-iteration
// One pass over the local pixels (no N x r matrix): per non-sample pixel i
//     k_i[s] = exp2(nsw*(dr^2+dc^2) + npw*dv^2),  d_i = k_i . w  (fp64),
//     y_i = 1 (COLSUM) or recip(d_i),  z[s] += k_i[s] * y_i.
// One wave owns 64 consecutive pixels at a time (lane = pixel); the p affinities of a
// pixel stay in registers between the dot product and the accumulation, sample data and w
// are wave-uniform (scalar loads).  The 64-lane sum of k[s]*y is a transpose-reduce after
// which lane l owns samples {l, l+64, ...}; partials are fp64 per wave, summed by
// k_reduce_partials in a fixed order.
template <int PP>
__global__ __launch_bounds__(256) void k_sink_pass(int mode, const float* __restrict__ lum, GridSpec gs,
                                                   const Sample4* __restrict__ samples,
                                                   const double* __restrict__ w, float nsw, float npw,
                                                   unsigned pix0, long long M, double eps,
                                                   double* __restrict__ ybuf, double* __restrict__ partial) {
    constexpr int P64 = (PP + 63) & ~63;
    constexpr int NJ = P64 / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * 4;
    const long long wv = (long long)blockIdx.x * 4 + wave;
    const long long ntiles = (M + 63) >> 6;
    double acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.0;

    for (long long tile = wv; tile < ntiles; tile += nwaves) {
        const long long li = tile * 64 + lane;
        const bool valid = li < M;
        const unsigned gi = pix0 + (unsigned)(valid ? li : M - 1);
        const unsigned row = gi / (unsigned)gs.W, col = gi - row * (unsigned)gs.W;
        const float pr = (float)row, pc = (float)col, px = lum[gi];
        const bool live = valid && !is_sample_pixel(gs, (int)row, (int)col);

        // The sample table and w are wave-uniform and loop-invariant; an opaque zero offset per
        // tile keeps hipcc from hoisting all 6*PP scalar loads out of the tile loop (which spills
        // ~900 SGPRs and re-reads them with v_readlane every iteration).
        int zoff;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
        const Sample4* __restrict__ sp = samples + zoff;
        const double* __restrict__ wp = w + zoff;
        float k[P64];
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int s = 0; s < PP; ++s) {
            const Sample4 sm = sp[s];
            const float kv = affinity_value(pr, pc, px, sm, nsw, npw);
            k[s] = kv;
            const double t = (double)kv * wp[s];
            if ((s & 3) == 0) d0 += t;
            else if ((s & 3) == 1) d1 += t;
            else if ((s & 3) == 2) d2 += t;
            else d3 += t;
        }
#pragma unroll
        for (int s = PP; s < P64; ++s) k[s] = 0.f;
        double y = 1.0;
        if (mode != ROWPASS_COLSUM) y = recip_or_zero_d((d0 + d1) + (d2 + d3), eps);
        if (!live) y = 0.0;
        if (ybuf != nullptr && valid) ybuf[li] = y;
        const float yf = (float)y;
#pragma unroll
        for (int s = 0; s < PP; ++s) k[s] *= yf;
        fold_all<P64, P64, 64>(k, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] += (double)k[j];
    }
    // combine the block's 4 waves (fixed order), one partial row per block
    __shared__ double sacc[3][P64];
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) sacc[wave - 1][64 * j + lane] = acc[j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            partial[(size_t)blockIdx.x * P64 + 64 * j + lane] =
                (acc[j] + sacc[0][64 * j + lane]) + (sacc[1][64 * j + lane] + sacc[2][64 * j + lane]);
    }
}

int sink_pass_ld(int p) { return (p + 63) & ~63; }
int sink_pass_max_p() { return 256; }

static int sink_grid(long long M) {
    const long long ntiles = (M + 63) / 64;
    long long g = (ntiles + 3) / 4;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}
int sink_pass_rows(long long M) { return sink_grid(M); }

hipError_t sink_pass(hipStream_t s, int mode, const float* d_lum, GridSpec gs, const Sample4* d_samples,
                     int p, const double* d_w, float nsw, float npw, long long pix0, long long M, double eps,
                     double* d_ybuf, double* d_partial) {
    const int pp = (p + 15) & ~15;
    const int grid = sink_grid(M);
#define NLE_SP_CASE(PPV)                                                                                  \
    case PPV:                                                                                             \
        hipLaunchKernelGGL((k_sink_pass<PPV>), dim3(grid), dim3(256), 0, s, mode, d_lum, gs, d_samples,  \
                           d_w, nsw, npw, (unsigned)pix0, M, eps, d_ybuf, d_partial);                     \
        break;
    switch (pp) {
        NLE_SP_CASE(16)
        NLE_SP_CASE(32)
        NLE_SP_CASE(48)
        NLE_SP_CASE(64)
        NLE_SP_CASE(80)
        NLE_SP_CASE(96)
        NLE_SP_CASE(112)
        NLE_SP_CASE(128)
        NLE_SP_CASE(144)
        NLE_SP_CASE(160)
        NLE_SP_CASE(176)
        NLE_SP_CASE(192)
        NLE_SP_CASE(208)
        NLE_SP_CASE(224)
        NLE_SP_CASE(240)
        NLE_SP_CASE(256)
        default:
            return hipErrorInvalidValue;
    }
#undef NLE_SP_CASE
    return hipGetLastError();
}

// ------------------------------------------------------------------ Sinkhorn update (p-, r-sized)
// ---- the p-sized update between two passes
// After the pass with scaling vector u (w = B u):  y_a = 1 or recip(V_A[a] . u) for the p sample pixels
// (exact fp64 rows, :275 top block), then, in the reference's own order of operations (:239,243),
//     t = B^T z + V_A^T y_A,   u' = lambda o t,   w' = B u',   s_A' = V_A u'
// (the passes only need w and the samples' row sums s_A).  The products stay FACTORED: two dependent
// matrix-vector launches, stage A (r outputs) and stage B (2p outputs).  Round 1 folded them into one matrix
// Mu = [B; V_A] diag(lambda) [B; V_A]^T; its upper left block is pinv(K_A) written out, entries ~1/lambda_min, and the
// rounding of those entries (eps / lambda_min, unstructured) is not damped by the k_i . v_small <= sqrt(lambda)
// factor that damps the rounding of the factored form.  On the README images, where lambda_min sits at the 1e-10
// cut, that cost 1e-4 ... 1e-3 per layer and flipped a rank decision; the factored form agrees with the oracle to
// 1e-7 there (tests/test_readme_pairs_gpu.py).
//   X1: 2p x r column-major (column k contiguous): [B; V_A]
//   X2: 2p x r row-major    (row o contiguous):    [B; V_A]
// Cholesky form (K_A full rank, V_A := L, B := L^-T, lambda := 1, r == p): the projector blocks are the identity and
// V_A diag(lambda) V_A^T is K_A exactly, used as such: X1 = [L^-T; 0], u' = L^-1 z, w' = L^-T u' + y_A,
// s_A' = z + K_A y_A -- the lower half of X2 then holds the rows of K_A and is contracted with y_A.
//
// stage A: v = [z; y_A] (every workgroup builds it in LDS; workgroup 0 also stores it), u'_k = lambda_k X1[:,k] . v
// z: zrows slices of stride zld (summed here in a fixed order); sA_cur: s_A of the scaling the pass used
// (ignored by the column-sum pass, where y_A = 1)
__global__ __launch_bounds__(256) void k_sink_update_a(int mode, int p, int r, const double* __restrict__ X1,
                                                       const double* __restrict__ lam, const double* __restrict__ z,
                                                       int zrows, int zld, const double* __restrict__ sA_cur, double eps,
                                                       double* __restrict__ v_out, double* __restrict__ u_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* v = reinterpret_cast<double*>(smem_raw);  // [2p] = [z; y_A]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n2 = 2 * p;
    for (int a = tid; a < p; a += 256) {
        double t = 0.0;
        for (int q = 0; q < zrows; ++q) t += z[(size_t)q * zld + a];
        const double ya = (mode == ROWPASS_COLSUM) ? 1.0 : recip_or_zero_d(sA_cur[a], eps);
        v[a] = t;
        v[p + a] = ya;
        if (blockIdx.x == 0) {
            v_out[a] = t;
            v_out[p + a] = ya;
        }
    }
    __syncthreads();
    const int k = blockIdx.x * 4 + wave;
    if (k >= r) return;  // wave-uniform
    const double* col = X1 + (size_t)k * n2;
    double s0 = 0.0, s1 = 0.0;
    int i = lane;
    for (; i + 64 < n2; i += 128) {
        s0 += col[i] * v[i];
        s1 += col[i + 64] * v[i + 64];
    }
    if (i < n2) s0 += col[i] * v[i];
    double sum = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) u_out[k] = lam[k] * sum;
}

// stage B: w'_o = X2[o,:] . u' (+ y_A[o] in the Cholesky form);  s_A'_a = X2[p+a,:] . u'  or  z_a + K_A[a,:] . y_A
__global__ __launch_bounds__(256) void k_sink_update_b(int p, int r, int chol, const double* __restrict__ X2,
                                                       const double* __restrict__ u, const double* __restrict__ v,
                                                       double* __restrict__ sA_next, double* __restrict__ w_next) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x * 4 + wave;
    if (o >= 2 * p) return;  // wave-uniform
    const bool exact = chol != 0 && o >= p;  // contract the row of K_A with y_A
    const double* row = X2 + (size_t)o * r;
    const double* x = exact ? v + p : u;
    double s0 = 0.0, s1 = 0.0;
    int i = lane;
    for (; i + 64 < r; i += 128) {
        s0 += row[i] * x[i];
        s1 += row[i + 64] * x[i + 64];
    }
    if (i < r) s0 += row[i] * x[i];
    double sum = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) {
        if (o < p) w_next[o] = chol ? sum + v[p + o] : sum;
        else sA_next[o - p] = chol ? v[o - p] + sum : sum;
    }
}

// d_v: 2p doubles, d_u: r doubles of scratch
hipError_t sink_update(hipStream_t s, int mode, int p, int r, bool chol, const double* d_X1, const double* d_X2,
                       const double* d_lam, const double* d_z, int zrows, int zld, const double* d_sA_cur, double eps,
                       double* d_v, double* d_u, double* d_sA_next, double* d_w_next) {
    const size_t shm = (size_t)2 * p * sizeof(double);
    hipLaunchKernelGGL(k_sink_update_a, dim3((unsigned)((r + 3) / 4)), dim3(256), shm, s, mode, p, r, d_X1, d_lam, d_z,
                       zrows, zld, d_sA_cur, eps, d_v, d_u);
    hipLaunchKernelGGL(k_sink_update_b, dim3((unsigned)((2 * p + 3) / 4)), dim3(256), 0, s, p, r, chol ? 1 : 0, d_X2, d_u,
                       d_v, d_sA_next, d_w_next);
    return hipGetLastError();
}

// ------------------------------------------------------------------ Gram in sample space (fp64 MFMA)
// Gk = sum over the local NON-sample pixels of c_i^2 k_i k_i^T (p x p), the N-sized part of
// Wab*Wab^T (:296).  In sample space this matrix needs ~1e-9 relative accuracy (DESIGN.md
// "Numerics"): products and sums run on v_mfma_f64_16x16x4_f64 (fp64 in, fp64 accumulate).
// grid.x = row chunks, grid.y = groups of 4*kG64TilesPerWave upper-triangular 16x16 tiles.
// Each stage the block generates 32 rows z_i = c_i k_i (fp64) into LDS: one wave per row, lane
// = sample, so the ds_write_b64 are contiguous; every lane keeps its <= 4 samples in registers.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kG64Rows = 32;

int gram64_ld(int p) { return (p + 15) & ~15; }
int gram64_num_tiles(int p) {
    const int nt = gram64_ld(p) / 16;
    return nt * (nt + 1) / 2;
}
static int gram64_chunk_rows(long long M) {
    long long fl = (M + 511) / 512;
    fl = ((fl + kG64Rows - 1) / kG64Rows) * kG64Rows;
    return (int)std::max<long long>(fl, kG64Rows);
}
static long long gram64_num_chunks(long long M) {
    const int fl = gram64_chunk_rows(M);
    return (M + fl - 1) / fl;
}
size_t gram64_partial_elems(long long M, int p) {
    return (size_t)gram64_num_chunks(M) * gram64_num_tiles(p) * 256;
}

template <int TPW>
__global__ __launch_bounds__(256) void k_gram64(const float* __restrict__ lum, GridSpec gs,
                                                const Sample4* __restrict__ samples, int p, int ld16, int ldz,
                                                float nsw, float npw, unsigned pix0, long long M,
                                                const double* __restrict__ cvec, int chunk_rows, int ntiles,
                                                double* __restrict__ partial) {
    constexpr int RB = kG64Rows, RW = RB / 4;  // rows per stage, rows generated per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sZ = reinterpret_cast<double*>(smem_raw);  // [RB][ldz]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int nt = ld16 >> 4;

    // tile j of this wave: byte offsets of its A / B column inside an LDS row.  Slots past the
    // tile list alias tile 0 (computed, never stored) so that the MFMA loop is branch-free.
    int offA[TPW], offB[TPW];
    const int tbase = (blockIdx.y * 4 + wave) * TPW;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        int t = tbase + j, ti = 0;
        if (t >= ntiles) t = 0;
        while (t >= nt - ti) {
            t -= nt - ti;
            ++ti;
        }
        offA[j] = (ti * 16 + l15) * 8;
        offB[j] = ((ti + t) * 16 + l15) * 8;
    }
    f64x4 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f64x4{0.0, 0.0, 0.0, 0.0};

    // this lane's samples: s = lane + 64 m
    Sample4 ms[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int sidx = lane + 64 * m;
        ms[m] = (sidx < p) ? samples[sidx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const long long c0 = (long long)blockIdx.x * chunk_rows;
    const long long c1 = min(M, c0 + (long long)chunk_rows);
    // lanes 0..RW-1 fetch (row, col, lum, c) of the RW rows this wave generates in a stage; the
    // fetch for stage n+1 is issued before the MFMA loop of stage n
    auto fetch = [&](long long rb, float& fr, float& fc, float& fx, double& fcf) {
        const long long li = rb + wave * RW + (lane & (RW - 1));
        const bool valid = li < c1;
        const unsigned gi = pix0 + (unsigned)(valid ? li : c1 - 1);
        const unsigned row = gi / (unsigned)gs.W, col = gi - row * (unsigned)gs.W;
        fr = (float)row;
        fc = (float)col;
        fx = lum[gi];
        fcf = (valid && !is_sample_pixel(gs, (int)row, (int)col)) ? cvec[li] : 0.0;
    };
    float q_r, q_c, q_x;
    double q_cf;
    fetch(c0, q_r, q_c, q_x, q_cf);
    for (long long rb = c0; rb < c1; rb += RB) {
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const float pr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_r), rr));
            const float pc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_c), rr));
            const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_x), rr));
            const int clo = __builtin_amdgcn_readlane((int)(__double_as_longlong(q_cf) & 0xffffffffll), rr);
            const int chi = __builtin_amdgcn_readlane((int)(__double_as_longlong(q_cf) >> 32), rr);
            const double cf = __longlong_as_double(((long long)chi << 32) | (unsigned)clo);
            double* zrow = sZ + (size_t)(wave * RW + rr) * ldz;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int sidx = lane + 64 * m;
                if (sidx < ld16)
                    zrow[sidx] = (sidx < p) ? cf * (double)affinity_value(pr, pc, px, ms[m], nsw, npw) : 0.0;
            }
        }
        if (rb + RB < c1) fetch(rb + RB, q_r, q_c, q_x, q_cf);
        __syncthreads();
        const char* zbase = reinterpret_cast<const char*>(sZ) + (size_t)kq * ldz * 8;
#pragma unroll
        for (int kk = 0; kk < RB; kk += 4) {
            const char* zr = zbase + (size_t)kk * ldz * 8;
            // all operand reads of the k-step first (2*TPW LDS reads in flight), then the MFMAs
            double av[TPW], bv[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                av[j] = *reinterpret_cast<const double*>(zr + offA[j]);
                bv[j] = *reinterpret_cast<const double*>(zr + offB[j]);
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (tbase + j < ntiles) {
            double* out = partial + ((size_t)blockIdx.x * ntiles + (tbase + j)) * 256;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(kq + 4 * e) * 16 + l15] = acc[j][e];
        }
    }
}

__global__ __launch_bounds__(256) void k_gram64_reduce(const double* __restrict__ partial, int nchunks, int ntiles,
                                                       double* __restrict__ tiles) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)ntiles * 256;
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

template <int TPW>
static hipError_t launch_gram64(hipStream_t s, dim3 grid, size_t shm, const float* d_lum, GridSpec gs,
                                const Sample4* d_samples, int p, int ld16, int ldz, float nsw, float npw,
                                long long pix0, long long M, const double* d_c, int fl, int ntiles,
                                double* d_partial) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram64<TPW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_gram64<TPW>), grid, dim3(256), shm, s, d_lum, gs, d_samples, p, ld16, ldz, nsw, npw,
                       (unsigned)pix0, M, d_c, fl, ntiles, d_partial);
    return hipGetLastError();
}

hipError_t gram64(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, int p, float nsw,
                  float npw, long long pix0, long long M, const double* d_c, double* d_partial, double* d_tiles) {
    if (p > 256) return hipErrorInvalidValue;
    const int ld16 = gram64_ld(p);
    const int ldz = ld16 + ((ld16 % 32 == 16) ? 0 : 16);  // row stride == 16 (mod 32) doubles: conflict-free b64 reads
    const int ntiles = gram64_num_tiles(p);
    const int fl = gram64_chunk_rows(M);
    const long long nchunks = gram64_num_chunks(M);
    const size_t shm = (size_t)kG64Rows * ldz * sizeof(double);
    // tiles per wave: the smallest of {4, 8, 12, 16, 20, 23} that covers the list with one group
    static const int kTpw[6] = {4, 8, 12, 16, 20, kG64TilesPerWave};
    int tpw = kG64TilesPerWave;
    for (int i = 0; i < 6; ++i)
        if (4 * kTpw[i] >= ntiles) {
            tpw = kTpw[i];
            break;
        }
    const int groups = (ntiles + 4 * tpw - 1) / (4 * tpw);
    const dim3 grid((unsigned)nchunks, (unsigned)groups);
    hipError_t e;
#define NLE_G64(T) e = launch_gram64<T>(s, grid, shm, d_lum, gs, d_samples, p, ld16, ldz, nsw, npw, pix0, M, d_c, fl, ntiles, d_partial)
    switch (tpw) {
        case 4: NLE_G64(4); break;
        case 8: NLE_G64(8); break;
        case 12: NLE_G64(12); break;
        case 16: NLE_G64(16); break;
        case 20: NLE_G64(20); break;
        default: NLE_G64(kG64TilesPerWave); break;
    }
#undef NLE_G64
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gram64_reduce, dim3((unsigned)ntiles), dim3(256), 0, s, d_partial, (int)nchunks, ntiles,
                       d_tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------ projection in sample space (fp64 MFMA)
// V_i = c_i k_i^T D  (D = P[:, :q] R_A T2, p x K', reference :327 in sample space): the Nystrom
// extension of the K' retained eigenvectors to every pixel.  D has entries of both signs up to
// ~1e3 that cancel in the product, so D stays fp64 and the contraction runs on
// v_mfma_f64_16x16x4_f64; the A operand (lane: pixel l&15, sample k0 + (l>>4)) is the affinity
// generated in registers.  Block = 4 waves x 32 pixels; D is staged through LDS in 32-sample chunks.
template <int NT>
__global__ __launch_bounds__(256) void k_project64(const float* __restrict__ lum, GridSpec gs,
                                                   const Sample4* __restrict__ samples, int p, float nsw,
                                                   float npw, unsigned pix0, long long M,
                                                   const double* __restrict__ Dm, int ldd,
                                                   const double* __restrict__ cvec, float* __restrict__ V, int ldv) {
    constexpr int KB = 32;
    constexpr int LDB = NT * 16 + ((NT & 1) ? 0 : 16);
    __shared__ __attribute__((aligned(16))) double sB[KB][LDB];
    __shared__ Sample4 sS[KB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const long long m0 = (long long)blockIdx.x * 128 + wave * 32;

    float pr[2], pc[2], px[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        long long li = m0 + mt * 16 + l15;
        if (li >= M) li = M - 1;
        const unsigned gi = pix0 + (unsigned)li;
        const unsigned row = gi / (unsigned)gs.W;
        pr[mt] = (float)row;
        pc[mt] = (float)(gi - row * (unsigned)gs.W);
        px[mt] = lum[gi];
    }
    f64x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[mt][n] = f64x4{0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < p; k0 += KB) {
        __syncthreads();
        for (int idx = tid; idx < KB * NT * 16; idx += 256) {
            const int kk = idx / (NT * 16), cc = idx % (NT * 16);
            sB[kk][cc] = (k0 + kk < p && cc < ldd) ? Dm[(size_t)(k0 + kk) * ldd + cc] : 0.0;
        }
        if (tid < KB) sS[tid] = (k0 + tid < p) ? samples[k0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < KB; kk += 4) {
            const Sample4 sm = sS[kk + kq];
            const double a0 = (double)affinity_value(pr[0], pc[0], px[0], sm, nsw, npw);
            const double a1 = (double)affinity_value(pr[1], pc[1], px[1], sm, nsw, npw);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const double bv = sB[kk + kq][n * 16 + l15];
                acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1][n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long li = m0 + mt * 16 + kq + 4 * e;
            if (li < M) {
                const double cf = cvec[li];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = n * 16 + l15;
                    if (col < ldv) V[(size_t)li * ldv + col] = (float)(cf * acc[mt][n][e]);
                }
            }
        }
}

__host__ __device__ constexpr int project64_res_ldb(int nt, int rem) {
    // smallest row stride >= nt*16 + rem that is == 16 (mod 32) doubles (conflict-free ds_read_b64 of 4 rows)
    int w = nt * 16 + rem;
    int ld = (w + 15) / 16 * 16;
    if (ld % 32 == 0) ld += 16;
    return ld;
}

// Resident-D variant: the whole D (p x NT*16 fp64) and the sample table stay in LDS, workgroups are
// persistent over 128-pixel tiles, so the MFMA loop runs without staging barriers.
template <int NT, int NW, int REM>
__global__ __launch_bounds__(NW * 64) void k_project64_res(const float* __restrict__ lum, GridSpec gs,
                                                           const Sample4* __restrict__ samples, int p, float nsw,
                                                           float npw, unsigned pix0, long long M,
                                                           const double* __restrict__ Dm, int ldd,
                                                           const double* __restrict__ cvec, float* __restrict__ V,
                                                           int ldv) {
    // LDS row: NT*16 MFMA columns [+ REM leftover columns handled on the VALU], stride == 16 (mod 32) doubles
    constexpr int LDB = project64_res_ldb(NT, REM);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int p4 = (p + 3) & ~3;
    double* sB = reinterpret_cast<double*>(smem_raw);                 // [p4][LDB]
    Sample4* sS = reinterpret_cast<Sample4*>(sB + (size_t)p4 * LDB);  // [p4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    constexpr int NCOL = NT * 16 + REM;
    for (int idx = tid; idx < p4 * NCOL; idx += NW * 64) {
        const int kk = idx / NCOL, cc = idx - kk * NCOL;
        sB[(size_t)kk * LDB + cc] = (kk < p && cc < ldd) ? Dm[(size_t)kk * ldd + cc] : 0.0;
    }
    for (int k = tid; k < p4; k += NW * 64) sS[k] = (k < p) ? samples[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const long long ntiles = (M + NW * 32 - 1) / (NW * 32);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * (NW * 32) + wave * 32;
        float pr[2], pc[2], px[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            long long li = m0 + mt * 16 + l15;
            if (li >= M) li = M - 1;
            const unsigned gi = pix0 + (unsigned)li;
            const unsigned row = gi / (unsigned)gs.W;
            pr[mt] = (float)row;
            pc[mt] = (float)(gi - row * (unsigned)gs.W);
            px[mt] = lum[gi];
        }
        f64x4 acc[2][NT];
        double rem0[REM > 0 ? REM : 1], rem1[REM > 0 ? REM : 1];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[mt][n] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < (REM > 0 ? REM : 1); ++q) rem0[q] = rem1[q] = 0.0;
        // software pipeline: the affinities of step k+1 are generated while the MFMAs of step k run
        Sample4 sm = sS[kq];
        double a0 = (double)affinity_value(pr[0], pc[0], px[0], sm, nsw, npw);
        double a1 = (double)affinity_value(pr[1], pc[1], px[1], sm, nsw, npw);
        for (int kk = 0; kk < p4; kk += 4) {
            const double* brow = sB + (size_t)(kk + kq) * LDB;
            double bv[NT], br[REM > 0 ? REM : 1];
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = brow[n * 16 + l15];
#pragma unroll
            for (int q = 0; q < REM; ++q) br[q] = brow[NT * 16 + q];
            const int kn = (kk + 4 < p4) ? kk + 4 : kk;
            const Sample4 sn = sS[kn + kq];
            const double c0 = a0, c1 = a1;
            a0 = (double)affinity_value(pr[0], pc[0], px[0], sn, nsw, npw);
            a1 = (double)affinity_value(pr[1], pc[1], px[1], sn, nsw, npw);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, bv[n], acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, bv[n], acc[1][n], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < REM; ++q) {  // leftover columns: this lane's (pixel l15, sample kk+kq) term
                rem0[q] += c0 * br[q];
                rem1[q] += c1 * br[q];
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long li = m0 + mt * 16 + kq + 4 * e;
                if (li < M) {
                    const double cf = cvec[li];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int col = n * 16 + l15;
                        if (col < ldv) V[(size_t)li * ldv + col] = (float)(cf * acc[mt][n][e]);
                    }
                }
            }
        if constexpr (REM > 0) {
#pragma unroll
            for (int q = 0; q < REM; ++q) {  // sum the 4 sample quarters (lanes l15 + 16 kq)
                rem0[q] += __shfl_xor(rem0[q], 16);
                rem0[q] += __shfl_xor(rem0[q], 32);
                rem1[q] += __shfl_xor(rem1[q], 16);
                rem1[q] += __shfl_xor(rem1[q], 32);
            }
            if (kq == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const long long li = m0 + mt * 16 + l15;
                    if (li < M) {
                        const double cf = cvec[li];
#pragma unroll
                        for (int q = 0; q < REM; ++q) {
                            const int col = NT * 16 + q;
                            if (col < ldv) V[(size_t)li * ldv + col] = (float)(cf * (mt == 0 ? rem0[q] : rem1[q]));
                        }
                    }
                }
            }
        }
    }
}

template <int NT, int NW, int REM>
static hipError_t launch_project64_res(hipStream_t s, long long M, const float* d_lum, GridSpec gs,
                                       const Sample4* d_samples, int p, float nsw, float npw, long long pix0,
                                       const double* d_D, int ldd, const double* d_c, float* d_V, int ldv) {
    const int p4 = (p + 3) & ~3;
    const size_t shm = (size_t)p4 * project64_res_ldb(NT, REM) * sizeof(double) + (size_t)p4 * sizeof(Sample4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_project64_res<NT, NW, REM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    const long long ntiles = (M + NW * 32 - 1) / (NW * 32);
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 256);
    hipLaunchKernelGGL((k_project64_res<NT, NW, REM>), dim3(grid), dim3(NW * 64), shm, s, d_lum, gs, d_samples, p, nsw,
                       npw, (unsigned)pix0, M, d_D, ldd, d_c, d_V, ldv);
    return hipGetLastError();
}

int project64_ld(int K) { return (K + 15) & ~15; }

hipError_t project64(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, int p, float nsw,
                     float npw, long long pix0, long long M, const double* d_D, int K, const double* d_c, float* d_V,
                     int ldv) {
    if (M <= 0) return hipSuccess;
    const int ldd = project64_ld(K);
    const int nt = ldd / 16;
    {   // D resident in LDS when it fits (one persistent 16-wave workgroup per CU); up to 4 leftover
        // columns beyond a multiple of 16 are done on the VALU instead of paying a whole MFMA tile
        int nt_r = K / 16, rem = K % 16;
        if (rem > 4 || nt_r == 0) {
            nt_r = (K + 15) / 16;
            rem = 0;
        }
        const int p4 = (p + 3) & ~3;
        const size_t shm = (size_t)p4 * project64_res_ldb(nt_r, rem) * sizeof(double) + (size_t)p4 * sizeof(Sample4);
        if (nt_r <= 4 && shm <= 150 * 1024 && std::getenv("NLE_PROJECT_CHUNKED") == nullptr) {
#define NLE_PR(NTV, REMV) \
    if (nt_r == NTV && rem == REMV) \
        return launch_project64_res<NTV, 16, REMV>(s, M, d_lum, gs, d_samples, p, nsw, npw, pix0, d_D, ldd, d_c, d_V, ldv);
            NLE_PR(1, 0) NLE_PR(1, 1) NLE_PR(1, 2) NLE_PR(1, 3) NLE_PR(1, 4)
            NLE_PR(2, 0) NLE_PR(2, 1) NLE_PR(2, 2) NLE_PR(2, 3) NLE_PR(2, 4)
            NLE_PR(3, 0) NLE_PR(3, 1) NLE_PR(3, 2) NLE_PR(3, 3) NLE_PR(3, 4)
            NLE_PR(4, 0) NLE_PR(4, 1) NLE_PR(4, 2) NLE_PR(4, 3) NLE_PR(4, 4)
#undef NLE_PR
        }
    }
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
#define NLE_PJ_CASE(NTV)                                                                                   \
    case NTV:                                                                                              \
        hipLaunchKernelGGL((k_project64<NTV>), grid, block, 0, s, d_lum, gs, d_samples, p, nsw, npw,      \
                           (unsigned)pix0, M, d_D, ldd, d_c, d_V, ldv);                                    \
        break;
    switch (nt) {
        NLE_PJ_CASE(1)
        NLE_PJ_CASE(2)
        NLE_PJ_CASE(3)
        NLE_PJ_CASE(4)
        NLE_PJ_CASE(5)
        NLE_PJ_CASE(6)
        NLE_PJ_CASE(7)
        NLE_PJ_CASE(8)
        default:
            return hipErrorInvalidValue;
    }
#undef NLE_PJ_CASE
    return hipGetLastError();
}


// ==================================================================== quantised-luminance fast path
// When the luminance plane is integer valued in [0, 255] -- it always is in the reference's
// pipeline, where it is the L channel of an 8-bit Lab image (src/filter.cpp:460-469) -- and
// because the sample set is a Cartesian grid (samplePixels, :56-80: sample s = (a, b), row a of
// nR, column b of nC), the affinity factorises into three table look-ups
//     k_i[s] = er[row_i][a] * ec[col_i][b] * Ep[x_i][s],
//     er[r][a] = exp(-(r - row_a)^2/hx^2), ec[c][b] = exp(-(c - col_b)^2/hx^2), Ep[x][s] = exp(-(x - y_s)^2/hy^2)
// (all fp64, exact integer arguments).  A Sinkhorn half-iteration for one image row r becomes
//     g[x][b]   = sum_a er[r][a] w[a,b] Ep[x][a,b]                       (256 x nC table, LDS)
//     d_i       = sum_b ec[c_i][b] g[x_i][b],  y_i = recip(d_i)          (nC fma per pixel)
//     h[x][b]  += ec[c_i][b] y_i                                          (LDS histogram)
//     z[a,b]   += er[r][a] sum_x Ep[x][a,b] h[x][b]
// i.e. 2 nC multiply-adds per pixel instead of p exponentials and 2p fp64 fma, in fp64 throughout.
namespace {
constexpr int kLevels = 256;
}

int sink_hist_max_cols() { return 36; }  // 2 * 256 * nC doubles of LDS

// ---- wave-level pre-reduction for the LDS histograms
// Flat image regions put many lanes of a wave on ONE histogram level, and same-address LDS atomics
// serialise (a flat row ran the pass 4.5x slower than a noisy one).  Before the atomics, up to
// kGroupRounds levels that at least kGroupMin lanes of the wave share are summed across the wave on
// the VALU (DPP) and added once by lane 63; the remaining lanes use their own atomics.
constexpr int kGroupMin = 12;
constexpr int kGroupRounds = 3;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {  // lanes without a source (or masked rows) read 0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, valid in lane 63 (all lanes must be active)
__device__ __forceinline__ double wave_sum63(double v) {
    v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_f64<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
    v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
    return v;
}
// Calls grouped(level, mine) for each level handled by a wave sum (wave-uniform call, `mine` marks the
// lanes of that level) and returns whether this lane still has to add its own value.
template <class G>
__device__ __forceinline__ bool wave_group_levels(bool active, int x, G&& grouped) {
    bool pend = active, tried = false;
#pragma unroll 1
    for (int round = 0; round < kGroupRounds; ++round) {
        const unsigned long long cand = __ballot(pend && !tried);
        if (cand == 0) break;
        const int lx = __builtin_amdgcn_readlane(x, __ffsll((long long)cand) - 1);
        const bool mine = pend && x == lx;
        const int cnt = __popcll(__ballot(mine));
        if (cnt >= kGroupMin) {
            grouped(lx, mine);
            if (mine) pend = false;
        } else {
            if (cnt < 3) break;  // a noisy stretch: no point in trying further leaders
            if (mine) tried = true;
        }
    }
    return pend;
}

__global__ __launch_bounds__(256) void k_check_levels(const float* __restrict__ lum, long long n,
                                                      int* __restrict__ flag) {
    bool bad = false;
    unsigned tiles = 0;  // bit t: some pixel has a level in [16 t, 16 t + 16)
    auto look = [&](const float v) {
        const bool ok = v >= 0.f && v <= (float)(kLevels - 1) && v == floorf(v);
        bad = bad || !ok;
        if (ok) tiles |= 1u << ((int)v >> 4);
    };
    // 16-byte loads on the aligned body of the plane (4-byte loads ran this 67 MB read at 0.6 TB/s), scalars at both ends
    const long long head = min(n, (long long)((4 - ((reinterpret_cast<unsigned long long>(lum) >> 2) & 3)) & 3));
    const long long nv = (n - head) >> 2;
    const float4* body = reinterpret_cast<const float4*>(lum + head);
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long long)gridDim.x * blockDim.x;
    for (long long i = gtid; i < nv; i += gsz) {
        const float4 v = body[i];
        look(v.x);
        look(v.y);
        look(v.z);
        look(v.w);
    }
    for (long long i = gtid; i < head; i += gsz) look(lum[i]);
    for (long long i = head + 4 * nv + gtid; i < n; i += gsz) look(lum[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tiles |= __shfl_xor(tiles, off);
    const bool any_bad = __any(bad);  // a vote of the whole wave: taken before the lanes part ways
    // one atomic per workgroup (one per wave, 16k of them on the same word, cost more than the 67 MB read)
    __shared__ unsigned s_tiles[4];
    __shared__ int s_bad[4];
    if ((threadIdx.x & 63) == 0) {
        s_tiles[threadIdx.x >> 6] = tiles;
        s_bad[threadIdx.x >> 6] = any_bad ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) atomicOr(flag, 1);
        atomicOr(flag + 1, (int)(s_tiles[0] | s_tiles[1] | s_tiles[2] | s_tiles[3]));
    }
}

// d_flag: 2 ints.  [0] != 0: the plane is not integer valued in [0, 255]; [1]: which 16-level tiles occur (bit t)
hipError_t check_levels(hipStream_t s, const float* d_lum, long long n, int* d_flag) {
    hipError_t e = hipMemsetAsync(d_flag, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return e;
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_check_levels, dim3((unsigned)g), dim3(256), 0, s, d_lum, n, d_flag);
    return hipGetLastError();
}

// er: [nrows_local][nR], ecT: [nC][W], Ep: [256][p]
__global__ void k_hist_tables(GridSpec gs, const Sample4* __restrict__ samples, int p, double inv_hx2,
                              double inv_hy2, int row0, int nrows_local, double* __restrict__ er,
                              double* __restrict__ ecT, double* __restrict__ Ep) {
    const long long n_er = (long long)nrows_local * gs.nSelRows, n_ec = (long long)gs.nSelCols * gs.W,
                    n_ep = (long long)kLevels * p;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_er + n_ec + n_ep;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < n_er) {
            const int r = row0 + (int)(i / gs.nSelRows), a = (int)(i % gs.nSelRows);
            const double d = (double)(r - (gs.rowOff + a * gs.rowStep));
            er[i] = exp(-(d * d) * inv_hx2);
        } else if (i < n_er + n_ec) {
            const long long j = i - n_er;
            const int b = (int)(j / gs.W), c = (int)(j % gs.W);
            const double d = (double)(c - (gs.colOff + b * gs.colStep));
            ecT[j] = exp(-(d * d) * inv_hx2);
        } else {
            const long long j = i - n_er - n_ec;
            const int x = (int)(j / p), sidx = (int)(j % p);
            const double d = (double)x - (double)samples[sidx].z;
            Ep[j] = exp(-(d * d) * inv_hy2);
        }
    }
}

hipError_t hist_tables(hipStream_t s, GridSpec gs, const Sample4* d_samples, int p, double hx, double hy, int row0,
                       int nrows_local, double* d_er, double* d_ecT, double* d_Ep) {
    hipLaunchKernelGGL(k_hist_tables, dim3(256), dim3(256), 0, s, gs, d_samples, p, 1.0 / (hx * hx), 1.0 / (hy * hy),
                       row0, nrows_local, d_er, d_ecT, d_Ep);
    return hipGetLastError();
}

// one workgroup per local image row; partial: [nrows_local][ldp] doubles
__global__ __launch_bounds__(256) void k_sink_hist(int mode, const float* __restrict__ lum, GridSpec gs, int p,
                                                   int ldp, int row0, const double* __restrict__ er,
                                                   const double* __restrict__ ecT, const double* __restrict__ Ep,
                                                   const double* __restrict__ w, double eps,
                                                   double* __restrict__ ybuf, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nC = gs.nSelCols, nR = gs.nSelRows, W = gs.W;
    double* g = reinterpret_cast<double*>(smem_raw);  // [256][nC]
    double* h = g + kLevels * nC;                      // [256][nC]
    double* ew = h + kLevels * nC;                     // [p]   er[r][a] * w[a,b]
    const int tid = threadIdx.x;
    const int lrow = blockIdx.x, r = row0 + lrow;
    const double* er_r = er + (size_t)lrow * nR;
    for (int i = tid; i < kLevels * nC; i += 256) h[i] = 0.0;
    if (mode != ROWPASS_COLSUM) {
        for (int sidx = tid; sidx < p; sidx += 256) ew[sidx] = er_r[sidx / nC] * w[sidx];
        __syncthreads();
        for (int i = tid; i < kLevels * nC; i += 256) {
            const int x = i / nC, b = i - x * nC;
            const double* ep = Ep + (size_t)x * p + b;
            double s0 = 0.0, s1 = 0.0;
            int a = 0;
            for (; a + 1 < nR; a += 2) {
                s0 += ew[a * nC + b] * ep[a * nC];
                s1 += ew[(a + 1) * nC + b] * ep[(a + 1) * nC];
            }
            if (a < nR) s0 += ew[a * nC + b] * ep[a * nC];
            g[i] = s0 + s1;
        }
    }
    __syncthreads();
    // is this image row a sample row?
    const int dr = r - gs.rowOff;
    const bool sample_row = dr >= 0 && (dr % gs.rowStep) == 0 && (dr / gs.rowStep) < nR;
    for (int c = tid; c < W; c += 256) {
        const size_t gi = (size_t)r * W + c;
        const int x = (int)lum[gi];
        bool smp = false;
        if (sample_row) {
            const int dc = c - gs.colOff;
            smp = dc >= 0 && (dc % gs.colStep) == 0 && (dc / gs.colStep) < nC;
        }
        double y = 1.0;
        if (mode != ROWPASS_COLSUM) {
            double s0 = 0.0, s1 = 0.0;
            int b = 0;
            for (; b + 1 < nC; b += 2) {
                s0 += ecT[(size_t)b * W + c] * g[x * nC + b];
                s1 += ecT[(size_t)(b + 1) * W + c] * g[x * nC + b + 1];
            }
            if (b < nC) s0 += ecT[(size_t)b * W + c] * g[x * nC + b];
            y = recip_or_zero_d(s0 + s1, eps);
        }
        if (smp) y = 0.0;
        if (ybuf != nullptr) ybuf[(size_t)lrow * W + c] = y;
        if (y != 0.0)
            for (int b = 0; b < nC; ++b) atomicAdd(&h[x * nC + b], ecT[(size_t)b * W + c] * y);
    }
    __syncthreads();
    for (int sidx = tid; sidx < ldp; sidx += 256) {
        double out = 0.0;
        if (sidx < p) {
            const int a = sidx / nC, b = sidx - a * nC;
            double s0 = 0.0, s1 = 0.0;
            for (int x = 0; x < kLevels; x += 2) {
                s0 += Ep[(size_t)x * p + sidx] * h[x * nC + b];
                s1 += Ep[(size_t)(x + 1) * p + sidx] * h[(x + 1) * nC + b];
            }
            out = er_r[a] * (s0 + s1);
        }
        partial[(size_t)lrow * ldp + sidx] = out;
    }
}

hipError_t sink_hist(hipStream_t s, int mode, const float* d_lum, GridSpec gs, int p, int ldp, int row0,
                     int nrows_local, const double* d_er, const double* d_ecT, const double* d_Ep,
                     const double* d_w, double eps, double* d_ybuf, double* d_partial) {
    if (nrows_local <= 0) return hipSuccess;
    const size_t shm = ((size_t)2 * kLevels * gs.nSelCols + p) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sink_hist),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sink_hist, dim3((unsigned)nrows_local), dim3(256), shm, s, mode, d_lum, gs, p, ldp, row0, d_er,
                       d_ecT, d_Ep, d_w, eps, d_ybuf, d_partial);
    return hipGetLastError();
}

// -------------------------------------------------------------------- tiled form of the table pass
// k_sink_hist re-reads the whole Ep table (256 x p doubles) twice per image row from L2, which is
// what bounds it.  The tiled form splits the pass into three kernels so that Ep is read ~once:
//   k_hist_g   : g[r][b,x] = sum_a er[r][a] w[a,b] Ep[x][a,b]   (fp64 MFMA; table columns are b-major)
//   k_hist_pix : per image row: d_i, y_i, h[r][x,b] += ec y      (g row and h row in LDS)
//   k_hist_hh  : HH[slab][x,b][a] = sum_{r in slab} er[r][a] h[r][x,b]
//   k_hist_z   : z[a,b] = sum_x Ep[x][a,b] sum_slab HH[slab][x,b][a]
// G (nrows x 256 nC) = er (nrows x nR) * WE (nR x 256 nC), WE[a][x,b] = w[a,b] Ep[x][a,b], on the fp64
// MFMA: a wave keeps the WE operands of its 16 columns in registers (nR <= 32: 8 k-steps of 4) and walks
// down `tiles_per_wave` 16-row tiles -- five loads, five MFMAs and one 16 x 16 store per tile at cfg4, so
// the kernel runs at the speed of its 8 B/element output stream.
// lev_t0, lev_nt: the 16-level tiles [lev_t0, lev_t0 + lev_nt) that occur in the image -- columns of other levels are never
// read by anybody and are not made.
__global__ __launch_bounds__(256) void k_hist_g(GridSpec gs, int p, int nrows, int tiles_per_wave, int lev_t0, int lev_nt,
                                                const double* __restrict__ er, const double* __restrict__ Ep,
                                                const double* __restrict__ w, double* __restrict__ g) {
    // the er rows of this workgroup's row range, staged once (coalesced) -- a per-tile global load of the A operand put
    // one memory latency on every 16-row tile, which is what bounded the kernel (25 us for an 84 MB stream at cfg4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sEr = reinterpret_cast<double*>(smem_raw);  // [tiles_per_wave * 16][nR]
    const int nC = gs.nSelCols, nR = gs.nSelRows, n = kLevels * nC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int t0 = blockIdx.y * tiles_per_wave, ntiles = (nrows + 15) >> 4;
    const int rbase = t0 * 16, rcount = min(nrows - rbase, tiles_per_wave * 16);
    for (int i = tid; i < rcount * nR; i += 256) sEr[i] = er[(size_t)rbase * nR + i];
    const int q = blockIdx.x * 4 + wave;  // column tile: sample column b = q / lev_nt, level tile lev_t0 + q % lev_nt
    const bool col_ok = q < nC * lev_nt;
    const int b = col_ok ? q / lev_nt : 0, x = (lev_t0 + (col_ok ? q % lev_nt : 0)) * 16 + l15;
    const int col = b * kLevels + x;  // table columns are b-major: col = b*256 + x
    double bop[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int a = ks * 4 + kq;
        bop[ks] = (col_ok && a < nR) ? w[a * nC + b] * Ep[(size_t)x * p + a * nC + b] : 0.0;
    }
    __syncthreads();
    if (!col_ok) return;  // wave-uniform (after the barrier)
    const int ksteps = (nR + 3) >> 2;
    for (int t = t0; t < min(ntiles, t0 + tiles_per_wave); ++t) {
        const int rl = (t - t0) * 16 + l15;  // row within the staged range
        double aop[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int a = ks * 4 + kq;
            aop[ks] = (ks < ksteps && rl < rcount && a < nR) ? sEr[rl * nR + a] : 0.0;
        }
        f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ks], bop[ks], acc, 0, 0, 0);
        if (col_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ro = t * 16 + kq + 4 * e;
                if (ro < nrows) g[(size_t)ro * n + col] = acc[e];
            }
        }
    }
}

static hipError_t launch_hist_g(hipStream_t s, GridSpec gs, int p, int nrows_local, const double* d_er, const double* d_Ep,
                                const double* d_w, double* d_g, int lev_t0 = 0, int lev_nt = kLevels / 16) {
    const int ntiles = (nrows_local + 15) / 16;
    const int gx = (gs.nSelCols * lev_nt + 3) / 4;
    // ~2560 waves on the chip (or one row tile per wave if the slab is short); at most 16 tiles (256 rows x nR <= 32
    // doubles = 64 KB of LDS) per workgroup
    const int cap = gs.nSelRows > 24 ? 8 : 16;  // the staged er rows: <= 32 KB of LDS per workgroup (cfg5: -3 % on the Sinkhorn stage)
    const int chunks = std::max(1, std::min(ntiles, std::max((640 + gx - 1) / gx, (ntiles + cap - 1) / cap)));
    const int tpw = (ntiles + chunks - 1) / chunks;
    const size_t shm = (size_t)tpw * 16 * gs.nSelRows * sizeof(double);
    if (shm > 48 * 1024) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_g), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)shm);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(k_hist_g, dim3((unsigned)gx, (unsigned)((ntiles + tpw - 1) / tpw)), dim3(256), shm, s, gs, p,
                       nrows_local, tpw, lev_t0, lev_nt, d_er, d_Ep, d_w, d_g);
    return hipGetLastError();
}

constexpr int kPixThreads = 512;
template <int NC>  // NC = nSelCols: compile-time so that the per-pixel loops carry no branches
__global__ __launch_bounds__(kPixThreads) void k_hist_pix(int mode, const float* __restrict__ lum, GridSpec gs, int row0,
                                                  const double* __restrict__ ecT, const double* __restrict__ g,
                                                  double eps, double* __restrict__ ybuf, double* __restrict__ hout,
                                                  const double* __restrict__ cvec, const float* __restrict__ xvec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int n = kLevels * NC;
    constexpr int NS = NC | 1;  // odd row stride (in doubles): spreads the levels of a wave over the LDS banks
    const int W = gs.W;
    double* sg = reinterpret_cast<double*>(smem_raw);  // [256][NS]
    double* sh = sg + kLevels * NS;                     // [256][NS]
    const int tid = threadIdx.x, lrow = blockIdx.x, r = row0 + lrow;
    const double* grow = g + (size_t)lrow * n;
    for (int i = tid; i < n; i += kPixThreads) {
        const int bb = i / kLevels, xx = i & (kLevels - 1);  // global tables are b-major, the LDS copies level-major
        sh[xx * NS + bb] = 0.0;
        sg[xx * NS + bb] = (mode == ROWPASS_RECIP) ? grow[i] : 0.0;
    }
    __syncthreads();
    const int dr = r - gs.rowOff;
    const bool sample_row = dr >= 0 && (dr % gs.rowStep) == 0 && (dr / gs.rowStep) < gs.nSelRows;
    for (int c0 = 0; c0 < W; c0 += kPixThreads) {  // wave-uniform trip count: the body uses cross-lane sums
        const bool inside = c0 + tid < W;
        const int c = inside ? c0 + tid : W - 1;
        const int x = (int)lum[(size_t)r * W + c];
        bool smp = !inside;
        if (sample_row) {
            const int dc = c - gs.colOff;
            smp = smp || (dc >= 0 && (dc % gs.colStep) == 0 && (dc / gs.colStep) < NC);
        }
        double e[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) e[b] = ecT[(size_t)b * W + c];
        double y = 1.0;
        if (mode == ROWPASS_XVEC) {  // apply: y_i = c_i x_i (c is 0 at sample pixels)
            y = cvec[(size_t)lrow * W + c] * (double)xvec[(size_t)r * W + c];
        } else if (mode == ROWPASS_RECIP) {
            double gv[NC];
#pragma unroll
            for (int b = 0; b < NC; ++b) gv[b] = sg[x * NS + b];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int b = 0; b < NC; ++b) {
                if (b & 1) s1 += e[b] * gv[b];
                else s0 += e[b] * gv[b];
            }
            y = recip_or_zero_d(s0 + s1, eps);
        }
        if (smp) y = 0.0;
        if (ybuf != nullptr && inside) ybuf[(size_t)lrow * W + c] = y;
        const bool own = wave_group_levels(y != 0.0, x, [&](int lx, bool mine) {
#pragma unroll
            for (int b = 0; b < NC; ++b) {
                const double t = wave_sum63(mine ? e[b] * y : 0.0);
                if ((tid & 63) == 63) atomicAdd(&sh[lx * NS + b], t);
            }
        });
        if (own) {
#pragma unroll
            for (int b = 0; b < NC; ++b) atomicAdd(&sh[x * NS + b], e[b] * y);
        }
    }
    __syncthreads();
    double* hrow = hout + (size_t)lrow * n;
    for (int i = tid; i < n; i += kPixThreads) {
        const int bb = i / kLevels, xx = i & (kLevels - 1);
        hrow[i] = sh[xx * NS + bb];
    }
}

// apply, expand half: out[i] = (float)(c_i * sum_b ec[c_i][b] g_r[x_i][b]) with g built from w' = D (f o t)
// Up to kDotLayers layers per launch: the tables of the launch's layers sit side by side in LDS, so the pixel's
// level, its nC column factors and c_i are loaded once for all of them.
constexpr int kDotLayers = 4;
constexpr int kDotThreads = 512;
template <int NC>
__global__ __launch_bounds__(kDotThreads) void k_hist_dot(const float* __restrict__ lum, GridSpec gs, int row0,
                                                          const double* __restrict__ ecT, const double* __restrict__ g,
                                                          size_t gstride, int nl, const double* __restrict__ cvec,
                                                          float* __restrict__ out, long long ostride, int round8) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int n = kLevels * NC;
    constexpr int NS = NC | 1;  // odd row stride, as in k_hist_pix
    constexpr int TS = kLevels * NS;
    const int W = gs.W;
    double* sg = reinterpret_cast<double*>(smem_raw);  // [nl][256][NS]
    const int tid = threadIdx.x, lrow = blockIdx.x, r = row0 + lrow;
    for (int l = 0; l < nl; ++l) {
        const double* grow = g + (size_t)l * gstride + (size_t)lrow * n;
        for (int i = tid; i < n; i += kDotThreads) sg[l * TS + (i & (kLevels - 1)) * NS + i / kLevels] = grow[i];  // b-major -> level-major
    }
    __syncthreads();
    for (int c = tid; c < W; c += kDotThreads) {
        const int x = (int)lum[(size_t)r * W + c];
        const double cv = cvec[(size_t)lrow * W + c];
        double e[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) e[b] = ecT[(size_t)b * W + c];
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            const double* t = sg + l * TS + x * NS;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int b = 0; b < NC; ++b) {
                if (b & 1) s1 += e[b] * t[b];
                else s0 += e[b] * t[b];
            }
            double v = cv * (s0 + s1);
            if (round8) v = rint(fmin(255.0, fmax(0.0, v)));  // src/filter.cpp:434-436 on the fp64 value (k_sorted_expand)
            out[(size_t)l * ostride + (size_t)lrow * W + c] = (float)v;
        }
    }
}

// p-, K-sized half of the sample-space apply (one workgroup):
//   t = D^T m + Vrows^T x_A,  W'[l] = D (resp_l o t),  YA[l][a] = Vrows[a] . (resp_l o t)
// D, Vrows: p x ldk row-major fp64; m: p column sums sum_i k_i c_i x_i; xA: x at the p sample pixels
__global__ __launch_bounds__(256) void k_apply_small(int p, int K, int ldk, int L, int ldw, const double* __restrict__ m,
                                                     const double* __restrict__ Dm, const double* __restrict__ Vrows,
                                                     const double* __restrict__ xA, const double* __restrict__ resp,
                                                     double* __restrict__ t_out, double* __restrict__ Wp,
                                                     double* __restrict__ YA) {
    // one workgroup per layer l = blockIdx.x; each recomputes t (K values, p terms each: cheaper than a second launch)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);  // [p]
    double* sx = sm + p;                                // [p]
    double* st = sx + p;                                // [K]
    double* sp = st + K;                                // [G][KP] partial sums of t
    const int tid = threadIdx.x, l = blockIdx.x;
    const int KP = K <= 64 ? 64 : 128, G = 256 / KP;
    for (int a = tid; a < p; a += 256) {
        sm[a] = m[a];
        sx[a] = xA[a];
    }
    __syncthreads();
    {
        const int k = tid % KP, g = tid / KP;
        double s0 = 0.0, s1 = 0.0;
        if (k < K)
            for (int a = g; a < p; a += G) {
                s0 += Dm[(size_t)a * ldk + k] * sm[a];
                s1 += Vrows[(size_t)a * ldk + k] * sx[a];
            }
        sp[g * KP + k] = s0 + s1;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        double s = 0.0;
        for (int g = 0; g < G; ++g) s += sp[g * KP + k];  // fixed order
        st[k] = s;
        if (l == 0) t_out[k] = s;
    }
    __syncthreads();
    const double* rl = resp + (size_t)l * K;
    for (int a = tid; a < ldw; a += 256) {
        double w = 0.0, ya = 0.0;
        if (a < p) {
            for (int k = 0; k < K; ++k) {
                const double gk = rl[k] * st[k];
                w += Dm[(size_t)a * ldk + k] * gk;
                ya += Vrows[(size_t)a * ldk + k] * gk;
            }
            YA[(size_t)l * p + a] = ya;
        }
        Wp[(size_t)l * ldw + a] = w;
    }
}

// Y[l][loc[a]] = YA[l][a] for the samples this rank owns (loc < 0: not local)
__global__ void k_scatter_samples(int p, int L, const long long* __restrict__ loc, const double* __restrict__ YA,
                                  float* __restrict__ Y, long long ystride, int round8) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= L * p) return;
    const int l = o / p, a = o - l * p;
    double v = YA[o];
    if (round8) v = rint(fmin(255.0, fmax(0.0, v)));
    if (loc[a] >= 0) Y[(size_t)l * ystride + loc[a]] = (float)v;
}

// HH[slab][a][col] = sum over the slab's image rows of er[r][a] h[r][col], col = x*nC + b: per slab a
// (nR x slab_rows) x (slab_rows x 256 nC) product on the fp64 MFMA.  One wave per 16 columns and slab; the
// slab's 16 row loads of a lane are independent, so they are all in flight together.
// grid (ceil(16 nC / 4), nslabs), slab_rows == 64.
// Epilogue: instead of storing the slab's HH tile (and re-reading all of them in a k_hist_z pass), the wave contracts its
// 16 levels with Ep on the spot: zpart[slab][x tile][a, b] = sum_{x in tile} Ep[x][a, b] HH[a][b, x]; a fixed-order reduce
// over the slabs and the 16 level tiles (reduce_partials) then gives z.  26 MB of HH written and read per pass become
// < 1 MB of partials, and one launch goes away.
__global__ __launch_bounds__(256) void k_hist_hh(int nC, int nR, int nrows, int slab_rows, int lev_t0, int lev_nt,
                                                 const double* __restrict__ er, const double* __restrict__ h,
                                                 const double* __restrict__ Ep, int p, int ldp, double* __restrict__ zpart) {
    // the slab's er rows, staged once for the four waves (they were re-read from global memory inside the MFMA loop: a
    // second dependent latency per 32 rows), and all of a lane's h loads of a 64-row half slab in flight together
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sEr = reinterpret_cast<double*>(smem_raw);  // [slab_rows][nR]
    const int n = kLevels * nC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.y * slab_rows, rcount = min(nrows - r0, slab_rows);
    for (int i = tid; i < rcount * nR; i += 256) sEr[i] = er[(size_t)r0 * nR + i];
    const int q = blockIdx.x * 4 + wave;  // column tile, as in k_hist_g: only the level tiles that occur
    const bool col_ok = q < nC * lev_nt;
    const int b = col_ok ? q / lev_nt : 0, xt = lev_t0 + (col_ok ? q % lev_nt : 0), x = xt * 16 + l15;
    const int col = b * kLevels + x;
    const bool two = nR > 16;
    f64x4 acc0 = f64x4{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
    __syncthreads();
    if (!col_ok) return;  // wave-uniform (after the barrier)
    for (int k0 = 0; k0 < slab_rows; k0 += 64) {
        double bop[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int rl = k0 + ks * 4 + kq;
            bop[ks] = (col_ok && rl < rcount) ? h[(size_t)(r0 + rl) * n + col] : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int rl = k0 + ks * 4 + kq;
            const bool rok = rl < rcount;
            const double a0 = (rok && l15 < nR) ? sEr[rl * nR + l15] : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bop[ks], acc0, 0, 0, 0);
            if (two) {
                const double a1 = (rok && 16 + l15 < nR) ? sEr[rl * nR + 16 + l15] : 0.0;
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bop[ks], acc1, 0, 0, 0);
            }
        }
    }
    // table columns are b-major (col = b * 256 + x) and a wave's 16 columns share b: lane (l15, kq) holds, for its level
    // x = x0 + l15, the sums of sample rows a = kq + 4 e (and 16 + kq + 4 e)
    double v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int a0 = kq + 4 * e, a1 = 16 + kq + 4 * e;
        v[e] = (col_ok && a0 < nR) ? acc0[e] * Ep[(size_t)x * p + a0 * nC + b] : 0.0;
        v[4 + e] = (two && col_ok && a1 < nR) ? acc1[e] * Ep[(size_t)x * p + a1 * nC + b] : 0.0;
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)  // sum over the 16 levels of the tile (lanes l15 of one kq group), fixed tree
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += __shfl_xor(v[e], off);
    if (l15 == 0 && col_ok) {
        double* out = zpart + ((size_t)blockIdx.y * lev_nt + (xt - lev_t0)) * ldp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int a0 = kq + 4 * e, a1 = 16 + kq + 4 * e;
            if (a0 < nR) out[a0 * nC + b] = v[e];
            if (two && a1 < nR) out[a1 * nC + b] = v[4 + e];
        }
    }
}

// z[s] = sum over the nparts (slab, level tile) partial rows, in a fixed order; columns s >= p come out as 0.
// 8 columns x 32 row groups per workgroup: ldp / 8 workgroups, each thread adds nparts / 32 values.
__global__ __launch_bounds__(256) void k_z_reduce(const double* __restrict__ zpart, int nparts, int p, int ldp,
                                                  double* __restrict__ z) {
    __shared__ double sm[32][8];
    const int c = threadIdx.x & 7, g = threadIdx.x >> 3, col = blockIdx.x * 8 + c;
    double s = 0.0;
    if (col < p)
        for (int r = g; r < nparts; r += 32) s += zpart[(size_t)r * ldp + col];
    sm[g][c] = s;
    __syncthreads();
    if (g == 0 && col < ldp) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sm[k][c];
        z[col] = t;
    }
}

int hist_tiled_max_rows_samples() { return 32; }
// image rows per slab of the HH stage: enough slabs to fill the chip's 1024 SIMDs about four times over with one wave
// per (16 table columns, slab), no more -- k_hist_z reads every slab's HH again (cfg4: 128 rows, 32 slabs, 5120 waves)
static int hist_slab_rows(GridSpec gs, int nrows_local) {
    const int ncoltiles = kLevels * gs.nSelCols / 16;
    const int want = std::max(1, (4096 + ncoltiles - 1) / ncoltiles);   // slabs wanted
    int rows = (nrows_local + want - 1) / want;
    rows = std::max(64, ((rows + 31) / 32) * 32);
    const int lds_rows = ((8192 / std::max(gs.nSelRows, 1)) / 32) * 32;  // k_hist_hh stages the slab's er rows: <= 64 KB
    return std::min(rows, std::max(64, lds_rows));
}
size_t hist_tiled_workspace_elems(GridSpec gs, int nrows_local) {
    const size_t n = (size_t)kLevels * gs.nSelCols;
    const int sr = hist_slab_rows(gs, nrows_local);
    const int nslabs = (nrows_local + sr - 1) / sr;
    const size_t ldp = ((size_t)gs.nSelRows * gs.nSelCols + 63) & ~(size_t)63;
    return 2 * (size_t)nrows_local * n + (size_t)nslabs * std::max(n * gs.nSelRows, (size_t)(kLevels / 16) * ldp);
}

// one Sinkhorn half-iteration, tiled form; d_ws: hist_tiled_workspace_elems doubles; d_z: ldp doubles
hipError_t sink_hist_tiled(hipStream_t s, int mode, const float* d_lum, GridSpec gs, int p, int ldp, int row0,
                           int nrows_local, const double* d_er, const double* d_ecT, const double* d_Ep,
                           const double* d_w, double eps, double* d_ybuf, double* d_ws, double* d_z,
                           LaunchObserver* obs, const double* d_cvec, const float* d_xvec, const SortedRows* sorted) {
    const int nC = gs.nSelCols, nR = gs.nSelRows;
    if (nC > 36 || nR > 32) return hipErrorInvalidValue;
    struct Scope {
        LaunchObserver* o;
        Scope(LaunchObserver* ob, int sub) : o(ob) { if (o) o->begin(sub); }
        ~Scope() { if (o) o->end(); }
    };
    const size_t n = (size_t)kLevels * nC;
    const int slab_rows = hist_slab_rows(gs, nrows_local), nslabs = (nrows_local + slab_rows - 1) / slab_rows;
    double* d_g = d_ws;
    double* d_h = d_g + (size_t)nrows_local * n;
    double* d_HH = d_h + (size_t)nrows_local * n;
    // the 16-level tiles that occur in the image (known with the sorted rows): the tables' other columns are neither made
    // (k_hist_g), stored (pass kernel) nor contracted (k_hist_hh)
    const int lev_t0 = sorted ? sorted->lev_t0 : 0, lev_nt = sorted ? sorted->lev_nt : kLevels / 16;
    if (mode == ROWPASS_RECIP) {
        Scope sc(obs, SUB_HIST_G);
        hipError_t eg = launch_hist_g(s, gs, p, nrows_local, d_er, d_Ep, d_w, d_g, lev_t0, lev_nt);
        if (eg != hipSuccess) return eg;
    }
    if (sorted != nullptr) {
        Scope sc(obs, SUB_HIST_PIX);
        hipError_t ep = sorted_pass(s, mode, gs, row0, nrows_local, sorted->scol, sorted->desc, sorted->first, sorted->E, d_g,
                                    eps, d_ybuf, d_h, d_cvec, d_xvec, sorted->rec, sorted->kappa, lev_t0, lev_nt, sorted->mom);
        if (ep != hipSuccess) return ep;
    } else {
        Scope sc(obs, SUB_HIST_PIX);
#define NLE_HP(NCV)                                                                                                 \
    case NCV:                                                                                                       \
        hipLaunchKernelGGL((k_hist_pix<NCV>), dim3((unsigned)nrows_local), dim3(kPixThreads),                      \
                           (size_t)2 * kLevels * ((NCV) | 1) * sizeof(double), s, mode,                            \
                           d_lum, gs, row0, d_ecT, d_g, eps, d_ybuf, d_h, d_cvec, d_xvec);                          \
        break;
        if (nC > 11) {
            hipError_t ea = hipSuccess;
            switch (nC) {  // > 64 KB of LDS: raise the limit of the instantiation that is about to run
#define NLE_HPA(NCV) case NCV: ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_pix<NCV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kLevels * ((NCV) | 1) * sizeof(double))); break;
                NLE_HPA(12) NLE_HPA(13) NLE_HPA(14) NLE_HPA(15) NLE_HPA(16) NLE_HPA(17) NLE_HPA(18) NLE_HPA(19) NLE_HPA(20)
                NLE_HPA(21) NLE_HPA(22) NLE_HPA(23) NLE_HPA(24) NLE_HPA(25) NLE_HPA(26) NLE_HPA(27) NLE_HPA(28) NLE_HPA(29)
                NLE_HPA(30) NLE_HPA(31) NLE_HPA(32) NLE_HPA(33) NLE_HPA(34) NLE_HPA(35) NLE_HPA(36)
#undef NLE_HPA
                default: break;
            }
            if (ea != hipSuccess) return ea;
        }
        switch (nC) {
            NLE_HP(1) NLE_HP(2) NLE_HP(3) NLE_HP(4) NLE_HP(5) NLE_HP(6) NLE_HP(7) NLE_HP(8) NLE_HP(9) NLE_HP(10) NLE_HP(11)
            NLE_HP(12) NLE_HP(13) NLE_HP(14) NLE_HP(15) NLE_HP(16) NLE_HP(17) NLE_HP(18) NLE_HP(19) NLE_HP(20)
            NLE_HP(21) NLE_HP(22) NLE_HP(23) NLE_HP(24) NLE_HP(25) NLE_HP(26) NLE_HP(27) NLE_HP(28) NLE_HP(29)
            NLE_HP(30) NLE_HP(31) NLE_HP(32) NLE_HP(33) NLE_HP(34) NLE_HP(35) NLE_HP(36)
            default: return hipErrorInvalidValue;
        }
#undef NLE_HP
    }
    Scope sc(obs, SUB_HIST_HH);
    {
        const size_t shm_hh = (size_t)slab_rows * nR * sizeof(double);  // slab_rows <= 1024 in practice; nR <= 32
        if (shm_hh > 48 * 1024) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_hh),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_hh);
            if (ea != hipSuccess) return ea;
        }
        hipLaunchKernelGGL(k_hist_hh, dim3((unsigned)((nC * lev_nt + 3) / 4), (unsigned)nslabs), dim3(256), shm_hh, s, nC, nR,
                           nrows_local, slab_rows, lev_t0, lev_nt, d_er, d_h, d_Ep, p, ldp, d_HH);
    }
    hipLaunchKernelGGL(k_z_reduce, dim3((unsigned)((ldp + 7) / 8)), dim3(256), 0, s, d_HH, nslabs * lev_nt, p, ldp, d_z);
    return hipGetLastError();
}

// layers of the sample-space apply's expand half that one k_hist_dot launch handles (LDS: one table each)
int apply_layers_per_launch(GridSpec gs) {
    const size_t table = (size_t)kLevels * (gs.nSelCols | 1) * sizeof(double);
    return (int)std::max<size_t>(1, std::min<size_t>(kDotLayers, (size_t)(144 * 1024) / table));
}

// expand half of the sample-space apply for `nl` <= apply_layers_per_launch layers: the g tables from the w' vectors
// (d_wl: nl vectors, stride ldw), then one dot kernel; d_ws: nl * nrows_local * 256 nC doubles;
// d_out: layer l at d_out + l * ostride
hipError_t apply_hist_layers(hipStream_t s, const float* d_lum, GridSpec gs, int p, int row0, int nrows_local,
                             const double* d_er, const double* d_ecT, const double* d_Ep, const double* d_wl, int ldw,
                             int nl, const double* d_c, double* d_ws, float* d_out, long long ostride,
                             LaunchObserver* obs, const SortedRows* sorted, bool round8) {
    const int nC = gs.nSelCols, nR = gs.nSelRows;
    if (nC > 36 || nR > 32 || nrows_local <= 0) return nrows_local <= 0 ? hipSuccess : hipErrorInvalidValue;
    const bool use_sorted = sorted != nullptr && nC <= sorted_expand_max_cols() && gs.W <= sorted_expand_max_width() &&
                            std::getenv("NLE_NO_SORTED_EXPAND") == nullptr;
    if (nl < 1 || nl > (use_sorted ? sorted_expand_layers(gs) : apply_layers_per_launch(gs))) return hipErrorInvalidValue;
    const size_t n = (size_t)kLevels * nC, gstride = (size_t)nrows_local * n;
    if (obs) obs->begin(SUB_HIST_G);
    for (int l = 0; l < nl; ++l) {
        hipError_t eg = launch_hist_g(s, gs, p, nrows_local, d_er, d_Ep, d_wl + (size_t)l * ldw, d_ws + (size_t)l * gstride,
                                      use_sorted ? sorted->lev_t0 : 0, use_sorted ? sorted->lev_nt : kLevels / 16);
        if (eg != hipSuccess) return eg;
    }
    if (obs) obs->end(), obs->begin(SUB_HIST_PIX);
    if (use_sorted) {
        hipError_t ex = sorted_expand(s, gs, nrows_local, sorted->scol, sorted->desc, sorted->E, d_ws, gstride, nl, d_c, d_out,
                                      ostride, sorted->rec, sorted->kappa, round8);
        if (obs) obs->end();
        return ex;
    }
#define NLE_HD(NCV)                                                                                                  \
    case NCV: {                                                                                                      \
        const size_t shm_d = (size_t)nl * kLevels * ((NCV) | 1) * sizeof(double);                                    \
        if (shm_d > 48 * 1024) {                                                                                     \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_dot<NCV>),                      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_d);             \
            if (ea != hipSuccess) return ea;                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_hist_dot<NCV>), dim3((unsigned)nrows_local), dim3(kDotThreads), shm_d, s, d_lum, gs,   \
                           row0, d_ecT, d_ws, gstride, nl, d_c, d_out, ostride, round8 ? 1 : 0);                     \
    } break;
    switch (nC) {
        NLE_HD(1) NLE_HD(2) NLE_HD(3) NLE_HD(4) NLE_HD(5) NLE_HD(6) NLE_HD(7) NLE_HD(8) NLE_HD(9) NLE_HD(10) NLE_HD(11)
        NLE_HD(12) NLE_HD(13) NLE_HD(14) NLE_HD(15) NLE_HD(16) NLE_HD(17) NLE_HD(18) NLE_HD(19) NLE_HD(20)
        NLE_HD(21) NLE_HD(22) NLE_HD(23) NLE_HD(24) NLE_HD(25) NLE_HD(26) NLE_HD(27) NLE_HD(28) NLE_HD(29)
        NLE_HD(30) NLE_HD(31) NLE_HD(32) NLE_HD(33) NLE_HD(34) NLE_HD(35) NLE_HD(36)
        default: return hipErrorInvalidValue;
    }
#undef NLE_HD
    if (obs) obs->end();
    return hipGetLastError();
}

hipError_t apply_small(hipStream_t s, int p, int K, int ldk, int L, int ldw, const double* d_m, const double* d_D,
                       const double* d_Vrows, const double* d_xA, const double* d_resp, double* d_t, double* d_Wp,
                       double* d_YA) {
    const size_t shm = (size_t)(2 * p + K + 256) * sizeof(double);
    hipLaunchKernelGGL(k_apply_small, dim3((unsigned)L), dim3(256), shm, s, p, K, ldk, L, ldw, d_m, d_D, d_Vrows, d_xA,
                       d_resp, d_t, d_Wp, d_YA);
    return hipGetLastError();
}

hipError_t scatter_samples(hipStream_t s, int p, int L, const long long* d_loc, const double* d_YA, float* d_Y,
                           long long ystride, bool round8) {
    hipLaunchKernelGGL(k_scatter_samples, dim3((unsigned)((L * p + 255) / 256)), dim3(256), 0, s, p, L, d_loc, d_YA, d_Y,
                       ystride, round8 ? 1 : 0);
    return hipGetLastError();
}

// -------------------------------------------------------------------- Gram via the same tables
// Gk[(a,b),(a',b')] = sum_r er[r][a] er[r][a'] sum_x Ep[x][a,b] Ep[x][a',b'] A_r[x][b,b'],
// A_r[x][b,b'] = sum over the non-sample pixels of image row r with level x of c^2 ec[c][b] ec[c][b'].
//   1. k_ghist_rows : A_r (256 x NP histogram in LDS, NP = nC(nC+1)/2 products per pixel) -> global
//   2. k_ghist_gemm : C[(a,a')][(b,b'),x] = sum_r EE[r][(a,a')] A_r[(b,b'),x]   (fp64 MFMA GEMM,
//                     M = nR(nR+1)/2, N = 256 NP, K = local image rows)
//   3. k_ghist_final: Gk[s][s'] = sum_x Ep[x][s] Ep[x][s'] C[(a,a')][x,(b,b')]
// ~NP LDS adds per pixel plus a 46 GFLOP GEMM at cfg4, instead of p^2/2 = 20 kFLOP per pixel.
int ghist_max_cols() { return 36; }  // one launch of k_ghist_rows up to 11, pair chunks beyond

__device__ __forceinline__ int tri_index(int i, int j, int n) {  // i <= j < n, row-major upper triangle
    return i * n - (i * (i - 1)) / 2 + (j - i);
}

// (the histogram takes most of a CU's LDS, so one workgroup per CU: 512 threads keep 8 waves on it)
constexpr int kGhistRowsThreads = 512;
__global__ __launch_bounds__(kGhistRowsThreads) void k_ghist_rows(const float* __restrict__ lum, GridSpec gs, int row0,
                                                    const double* __restrict__ ecT,
                                                    const double* __restrict__ cvec, double* __restrict__ Aout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* A = reinterpret_cast<double*>(smem_raw);  // [256][NP]
    const int nC = gs.nSelCols, W = gs.W, NP = nC * (nC + 1) / 2;
    const int tid = threadIdx.x, lrow = blockIdx.x, r = row0 + lrow;
    for (int i = tid; i < kLevels * NP; i += kGhistRowsThreads) A[i] = 0.0;
    __syncthreads();
    for (int c0 = 0; c0 < W; c0 += kGhistRowsThreads) {  // wave-uniform trip count: the body uses cross-lane sums
        const bool inside = c0 + tid < W;
        const int c = inside ? c0 + tid : W - 1;
        const double cf = inside ? cvec[(size_t)lrow * W + c] : 0.0;  // 0 at sample pixels
        const int x = (int)lum[(size_t)r * W + c];
        double q[11];
#pragma unroll
        for (int b = 0; b < 11; ++b) q[b] = (b < nC) ? cf * ecT[(size_t)b * W + c] : 0.0;
        const bool own = wave_group_levels(cf != 0.0, x, [&](int lx, bool mine) {
            double* Al = A + (size_t)lx * NP;
            int idx = 0;
#pragma unroll
            for (int b = 0; b < 11; ++b)
#pragma unroll
                for (int b2 = b; b2 < 11; ++b2)
                    if (b2 < nC) {
                        const double t = wave_sum63(mine ? q[b] * q[b2] : 0.0);
                        if ((tid & 63) == 63) atomicAdd(&Al[idx], t);
                        ++idx;
                    }
        });
        if (own) {
            double* Ax = A + (size_t)x * NP;
            int idx = 0;
#pragma unroll
            for (int b = 0; b < 11; ++b)
#pragma unroll
                for (int b2 = b; b2 < 11; ++b2)
                    if (b2 < nC) atomicAdd(&Ax[idx++], q[b] * q[b2]);
        }
    }
    __syncthreads();
    double* out = Aout + (size_t)lrow * kLevels * NP;  // global layout [pair][level]: k_ghist_final streams levels
    for (int i = tid; i < kLevels * NP; i += kGhistRowsThreads) out[i] = A[(i & (kLevels - 1)) * NP + i / kLevels];
}

// General form for 12 <= nC <= 36: the pair list does not fit in LDS at once, so a launch handles the
// pairs (b, b..nC-1) of sample columns b in [b0, b1) (<= kGhistChunkPairs pairs, chosen by the host) and
// writes that slice of A.
constexpr int kGhistMaxCols = 36;
constexpr int kGhistChunkPairs = 72;  // 256 * 72 * 8 B = 147 KB of LDS

__global__ __launch_bounds__(kGhistRowsThreads) void k_ghist_rows_chunk(const float* __restrict__ lum, GridSpec gs, int row0, int b0,
                                                          int b1, int pair_off, int npairs,
                                                          const double* __restrict__ ecT,
                                                          const double* __restrict__ cvec,
                                                          double* __restrict__ Aout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* A = reinterpret_cast<double*>(smem_raw);  // [256][npairs]
    const int nC = gs.nSelCols, W = gs.W, NP = nC * (nC + 1) / 2;
    const int tid = threadIdx.x, lrow = blockIdx.x, r = row0 + lrow;
    for (int i = tid; i < kLevels * npairs; i += kGhistRowsThreads) A[i] = 0.0;
    __syncthreads();
    for (int c0 = 0; c0 < W; c0 += kGhistRowsThreads) {  // wave-uniform trip count: the body uses cross-lane sums
        const bool inside = c0 + tid < W;
        const int c = inside ? c0 + tid : W - 1;
        const double cf = inside ? cvec[(size_t)lrow * W + c] : 0.0;  // 0 at sample pixels
        const int x = (int)lum[(size_t)r * W + c];
        double q[kGhistMaxCols];
#pragma unroll
        for (int b = 0; b < kGhistMaxCols; ++b) q[b] = (b < nC) ? cf * ecT[(size_t)b * W + c] : 0.0;
        // (no wave-level pre-reduction here: with 36 x 36 unrolled products it would double an already large
        // kernel and spill; flat regions cost this kernel their same-address serialisation)
        const bool own = cf != 0.0;
        if (own) {
            double* Ax = A + (size_t)x * npairs;
            int idx = 0;
#pragma unroll
            for (int b = 0; b < kGhistMaxCols; ++b) {
                if (b >= b0 && b < b1) {  // wave-uniform
#pragma unroll
                    for (int b2 = b; b2 < kGhistMaxCols; ++b2)
                        if (b2 < nC) atomicAdd(&Ax[idx++], q[b] * q[b2]);
                }
            }
        }
    }
    __syncthreads();
    double* out = Aout + (size_t)lrow * kLevels * NP + (size_t)pair_off * kLevels;  // [pair][level]
    for (int i = tid; i < kLevels * npairs; i += kGhistRowsThreads) {
        const int j = i / kLevels, x = i & (kLevels - 1);
        out[i] = A[x * npairs + j];
    }
}

// EE[r][m] = er[r][a] er[r][a'] for the m-th pair a <= a' (row stride ldm, zero padded)
__global__ void k_ghist_ee(const double* __restrict__ er, int nrows, int nR, int ldm, double* __restrict__ EE) {
    const long long n = (long long)nrows * ldm;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ldm), m = (int)(i % ldm);
        double v = 0.0;
        if (m < nR * (nR + 1) / 2) {
            int a = 0, t = m;
            while (t >= nR - a) {
                t -= nR - a;
                ++a;
            }
            v = er[(size_t)r * nR + a] * er[(size_t)r * nR + a + t];
        }
        EE[i] = v;
    }
}

// C[z] (ldm x N) = EE^T (rows [z ksplit, (z+1) ksplit) of K x ldm) * A (same rows of K x N); one 16-column
// tile per wave, MT m-tiles per wave.  The image rows are split over gridDim.z so that the launch fills the
// chip (N/64 blocks alone are fewer than the CUs); k_ghist_final adds the gridDim.z partial products in order.
template <int MT>
__global__ __launch_bounds__(256) void k_ghist_gemm(const double* __restrict__ EE, int ldm, const double* __restrict__ A,
                                                    long long N, int Ktot, int ksplit, double* __restrict__ Cz) {
    constexpr int KB = 16;
    constexpr int EPT = (KB * MT * 16 + 255) / 256;  // EE values staged per thread and k block
    __shared__ __attribute__((aligned(16))) double sE[KB][MT * 16 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const long long n0 = ((long long)blockIdx.x * 4 + wave) * 16;
    const int m0 = blockIdx.y * MT * 16;
    const bool ncol_ok = n0 + l15 < N;
    const int kbeg = blockIdx.z * ksplit, K = min(Ktot, kbeg + ksplit);
    double* C = Cz + (size_t)blockIdx.z * ldm * N;
    f64x4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = f64x4{0.0, 0.0, 0.0, 0.0};
    // software pipeline over the 16-row k blocks: the next block's EE slice and A operands are fetched into
    // registers while the MFMAs of the current one run; LDS is rewritten between two barriers
    double e_next[EPT], b_next[KB / 4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int idx = tid + q * 256, kk = idx / (MT * 16), mm = idx % (MT * 16);
            e_next[q] = (idx < KB * MT * 16 && k0 + kk < K && m0 + mm < ldm) ? EE[(size_t)(k0 + kk) * ldm + m0 + mm] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < KB / 4; ++q) {
            const int kr = k0 + 4 * q + kq;
            b_next[q] = (ncol_ok && kr < K) ? A[(size_t)kr * N + n0 + l15] : 0.0;
        }
    };
    if (kbeg < K) fetch(kbeg);
    for (int k0 = kbeg; k0 < K; k0 += KB) {
        __syncthreads();  // everybody is done reading the previous block's sE
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int idx = tid + q * 256;
            if (idx < KB * MT * 16) sE[idx / (MT * 16)][idx % (MT * 16)] = e_next[q];
        }
        double bcur[KB / 4];
#pragma unroll
        for (int q = 0; q < KB / 4; ++q) bcur[q] = b_next[q];
        __syncthreads();
        if (k0 + KB < K) fetch(k0 + KB);
#pragma unroll
        for (int q = 0; q < KB / 4; ++q) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(sE[4 * q + kq][j * 16 + l15], bcur[q], acc[j], 0, 0, 0);
        }
    }
    if (ncol_ok) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + j * 16 + kq + 4 * e;
                if (m < ldm) C[(size_t)m * N + n0 + l15] = acc[j][e];
            }
    }
}

// One wave per (row pair m = (a1 <= a2), column pair pc = (b1 <= b2)): its 256-level vector of C (summed over the
// GEMM's row splits in a fixed order) is read once, coalesced, and contracted with Ep for the one or two entries of
// Gk it feeds -- (a1,b1)x(a2,b2) and, when both pairs are off-diagonal, (a1,b2)x(a2,b1) -- plus their transposes.
__global__ __launch_bounds__(256) void k_ghist_final(const double* __restrict__ C, long long N, int nsplit, size_t zstride,
                                                     const double* __restrict__ Ep, int p, int nR, int nC,
                                                     double* __restrict__ Gk) {
    const int NP = nC * (nC + 1) / 2, NM = nR * (nR + 1) / 2;
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long long)NM * NP) return;  // wave-uniform
    const int m = (int)(w / NP), pc = (int)(w % NP);
    int a1 = 0, t = m;
    while (t >= nR - a1) {
        t -= nR - a1;
        ++a1;
    }
    const int a2 = a1 + t;
    int b1 = 0;
    t = pc;
    while (t >= nC - b1) {
        t -= nC - b1;
        ++b1;
    }
    const int b2 = b1 + t;
    const int s11 = a1 * nC + b1, s22 = a2 * nC + b2, s12 = a1 * nC + b2, s21 = a2 * nC + b1;
    const bool two = (a1 != a2) && (b1 != b2);
    const double* Cm = C + (size_t)m * N + (size_t)pc * kLevels;  // columns are [pair][level]
    double u = 0.0, v = 0.0;
#pragma unroll
    for (int j = 0; j < kLevels / 64; ++j) {
        const int x = j * 64 + lane;
        double cx = 0.0;
        for (int z = 0; z < nsplit; ++z) cx += Cm[z * zstride + x];  // fixed order
        const double* e = Ep + (size_t)x * p;
        u += e[s11] * e[s22] * cx;
        if (two) v += e[s12] * e[s21] * cx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u += __shfl_xor(u, off);
        v += __shfl_xor(v, off);
    }
    if (lane == 0) {
        Gk[(size_t)s11 * p + s22] = u;
        Gk[(size_t)s22 * p + s11] = u;
        if (two) {
            Gk[(size_t)s12 * p + s21] = v;
            Gk[(size_t)s21 * p + s12] = v;
        }
    }
}

// ---- Gram by index sums (sorted.hip: k_sorted_gsum has the derivation).  Rows: er[r][a] er[r][a'] =
// exp(-(a - a')^2 rs^2 / (2 hx^2)) F_{a+a'}(r), F_s(r) = exp(-2 (r - rowOff - s rs / 2)^2 / hx^2): the GEMM over image rows needs
// 2 nR - 1 rows instead of nR (nR + 1) / 2.
//   k_gsum_rowf : F[r][s] (row stride ldm, zero padded)
//   k_ghist_gemm: T[s][t, x] = sum_r F[r][s] S_r[t, x]
//   k_gsum_final: Gk[(a,b)][(a',b')] = kr_{|a-a'|} kc_{|b-b'|} sum_x Ep[x][a,b] Ep[x][a',b'] T[a+a'][b+b', x]
__global__ void k_gsum_rowf(GridSpec gs, int row0, int nrows, int ldm, double inv_hx2, double* __restrict__ F) {
    const long long n = (long long)nrows * ldm;
    const int ns = 2 * gs.nSelRows - 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ldm), sidx = (int)(i % ldm);
        double v = 0.0;
        if (sidx < ns) {
            const double d = (double)(row0 + r - gs.rowOff) - 0.5 * (double)sidx * (double)gs.rowStep;
            v = exp(-2.0 * d * d * inv_hx2);
        }
        F[i] = v;
    }
}

// one wave per pair of samples i <= j (grid: x = groups of 4 j's, y = i)
__global__ __launch_bounds__(256) void k_gsum_final(const double* __restrict__ T, long long N, int nsplit, size_t zstride,
                                                    const double* __restrict__ Ep, int p, int nC, double kr2, double kc2,
                                                    double* __restrict__ Gk) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.y, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j < i || j >= p) return;  // wave-uniform
    const int a1 = i / nC, b1 = i - a1 * nC, a2 = j / nC, b2 = j - a2 * nC;
    const int da = a1 - a2, db = b1 - b2;
    const double kappa = exp(-(double)(da * da) * kr2 - (double)(db * db) * kc2);
    const double* Tm = T + (size_t)(a1 + a2) * N + (size_t)(b1 + b2) * kLevels;
    double u = 0.0;
#pragma unroll
    for (int q = 0; q < kLevels / 64; ++q) {
        const int x = q * 64 + lane;
        double tx = 0.0;
        for (int z = 0; z < nsplit; ++z) tx += Tm[z * zstride + x];  // fixed order
        const double* e = Ep + (size_t)x * p;
        u += e[i] * e[j] * tx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) u += __shfl_xor(u, off);
    if (lane == 0) {
        u *= kappa;
        Gk[(size_t)i * p + j] = u;
        Gk[(size_t)j * p + i] = u;
    }
}

int ghist_ldm(int nR) { return ((nR * (nR + 1) / 2) + 15) & ~15; }
constexpr int kGhistMT = 14;
// image rows per GEMM split (a multiple of the 16-row k step) and the number of splits: ~4 workgroups per CU
static void ghist_split(long long N, int ldm, int nrows_local, int* ksplit, int* nsplit) {
    const long long base = ((N / 16 + 3) / 4) * ((ldm / 16 + kGhistMT - 1) / kGhistMT);
    int ns = (int)std::min<long long>(16, std::max<long long>(1, (1024 + base - 1) / base));
    int ks = (((nrows_local + ns - 1) / ns) + 15) & ~15;
    ks = std::max(ks, 16);
    *ksplit = ks;
    *nsplit = std::max(1, (nrows_local + ks - 1) / ks);
}
static int gsum_ldm(int nR) { return ((2 * nR - 1) + 15) & ~15; }
size_t ghist_workspace_elems(GridSpec gs, int nrows_local) {
    const size_t NP = (size_t)gs.nSelCols * (gs.nSelCols + 1) / 2, N = 256 * NP;
    const size_t ldm = (size_t)ghist_ldm(gs.nSelRows);
    int ks, ns;
    ghist_split((long long)N, (int)ldm, nrows_local, &ks, &ns);
    const size_t pairs = (size_t)nrows_local * N + (size_t)nrows_local * ldm + (size_t)ns * ldm * N;
    // the index-sum form (2 nC - 1 tables, 2 nR - 1 GEMM rows) needs less of each, but may split the rows further
    const size_t N2 = (size_t)256 * (2 * gs.nSelCols - 1), ldm2 = (size_t)gsum_ldm(gs.nSelRows);
    ghist_split((long long)N2, (int)ldm2, nrows_local, &ks, &ns);
    const size_t sums = (size_t)nrows_local * N2 + (size_t)nrows_local * ldm2 + (size_t)ns * ldm2 * N2;
    return std::max(pairs, sums);
}

// d_ws: ghist_workspace_elems doubles; d_Gk: p x p doubles (full symmetric matrix of this rank's rows)
hipError_t gram_hist(hipStream_t s, const float* d_lum, GridSpec gs, int p, int row0, int nrows_local,
                     const double* d_er, const double* d_ecT, const double* d_Ep, const double* d_c, double* d_ws,
                     double* d_Gk, LaunchObserver* obs, const SortedRows* sorted) {
    const int nC = gs.nSelCols, nR = gs.nSelRows;
    if (nC > kGhistMaxCols) return hipErrorInvalidValue;
    if (sorted != nullptr && sorted->E2 != nullptr) {  // index sums: 2 nC - 1 tables, 2 nR - 1 GEMM rows (sorted_gsum_ok)
        const long long N2 = (long long)kLevels * (2 * nC - 1);
        const int ldm2 = gsum_ldm(nR);
        double* d_S = d_ws;
        double* d_F = d_S + (size_t)nrows_local * N2;
        double* d_T = d_F + (size_t)nrows_local * ldm2;
        const double hx = sorted->hx;
        if (obs) obs->begin(SUB_GHIST_ROWS);
        hipError_t e2 = sorted_gram_sums(s, gs, nrows_local, sorted->scol, sorted->desc, sorted->first, sorted->E2, d_c, d_S, hx);
        if (e2 != hipSuccess) return e2;
        if (obs) obs->end(), obs->begin(SUB_GHIST_EE);
        hipLaunchKernelGGL(k_gsum_rowf, dim3(256), dim3(256), 0, s, gs, row0, nrows_local, ldm2, 1.0 / (hx * hx), d_F);
        constexpr int MT2 = 4;  // 2 nR - 1 <= 63: one group of four 16-row tiles
        int ksplit2, nsplit2;
        ghist_split(N2, ldm2, nrows_local, &ksplit2, &nsplit2);
        const dim3 grid2((unsigned)((N2 / 16 + 3) / 4), (unsigned)((ldm2 / 16 + MT2 - 1) / MT2), (unsigned)nsplit2);
        if (obs) obs->end(), obs->begin(SUB_GHIST_GEMM);
        hipLaunchKernelGGL((k_ghist_gemm<MT2>), grid2, dim3(256), 0, s, d_F, ldm2, d_S, N2, nrows_local, ksplit2, d_T);
        if (obs) obs->end(), obs->begin(SUB_GHIST_FINAL);
        const double rs = gs.rowStep, cs2 = gs.colStep;
        hipLaunchKernelGGL(k_gsum_final, dim3((unsigned)((p + 3) / 4), (unsigned)p), dim3(256), 0, s, d_T, N2, nsplit2,
                           (size_t)ldm2 * N2, d_Ep, p, nC, rs * rs / (2.0 * hx * hx), cs2 * cs2 / (2.0 * hx * hx), d_Gk);
        if (obs) obs->end();
        return hipGetLastError();
    }
    const int NP = nC * (nC + 1) / 2;
    const long long N = (long long)kLevels * NP;
    const int ldm = ghist_ldm(nR);
    double* d_A = d_ws;
    double* d_EE = d_A + (size_t)nrows_local * N;
    double* d_C = d_EE + (size_t)nrows_local * ldm;
    hipError_t e;
    if (obs) obs->begin(SUB_GHIST_ROWS);
    if (sorted != nullptr && nC <= sorted_gram_max_cols()) {
        e = sorted_gram_rows(s, gs, nrows_local, sorted->scol, sorted->desc, sorted->first, sorted->E, d_c, d_A, sorted->rec,
                             sorted->kappa);
        if (e != hipSuccess) return e;
    } else if (nC <= 11) {
        const size_t shm = (size_t)kLevels * NP * sizeof(double);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ghist_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ghist_rows, dim3((unsigned)nrows_local), dim3(kGhistRowsThreads), shm, s, d_lum, gs, row0,
                           d_ecT, d_c, d_A);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ghist_rows_chunk),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLevels * kGhistChunkPairs * (int)sizeof(double));
        if (e != hipSuccess) return e;
        int b0 = 0, off = 0;
        while (b0 < nC) {  // greedy: as many whole rows of the pair triangle as fit
            int b1 = b0, np = 0;
            while (b1 < nC && np + (nC - b1) <= kGhistChunkPairs) {
                np += nC - b1;
                ++b1;
            }
            if (b1 == b0) return hipErrorInvalidValue;  // cannot happen for nC <= 36 < 72
            hipLaunchKernelGGL(k_ghist_rows_chunk, dim3((unsigned)nrows_local), dim3(kGhistRowsThreads),
                               (size_t)kLevels * np * sizeof(double), s, d_lum, gs, row0, b0, b1, off, np, d_ecT, d_c, d_A);
            off += np;
            b0 = b1;
        }
    }
    if (obs) obs->end(), obs->begin(SUB_GHIST_EE);
    hipLaunchKernelGGL(k_ghist_ee, dim3(512), dim3(256), 0, s, d_er, nrows_local, nR, ldm, d_EE);
    constexpr int MT = kGhistMT;
    int ksplit, nsplit;
    ghist_split(N, ldm, nrows_local, &ksplit, &nsplit);
    const dim3 grid((unsigned)((N / 16 + 3) / 4), (unsigned)((ldm / 16 + MT - 1) / MT), (unsigned)nsplit);
    if (obs) obs->end(), obs->begin(SUB_GHIST_GEMM);
    hipLaunchKernelGGL((k_ghist_gemm<MT>), grid, dim3(256), 0, s, d_EE, ldm, d_A, N, nrows_local, ksplit, d_C);
    if (obs) obs->end(), obs->begin(SUB_GHIST_FINAL);
    const long long nwaves = (long long)(nR * (nR + 1) / 2) * NP;
    hipLaunchKernelGGL(k_ghist_final, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, d_C, N, nsplit,
                       (size_t)ldm * N, d_Ep, p, nR, nC, d_Gk);
    if (obs) obs->end();
    return hipGetLastError();
}

// -------------------------------------------------------------------- projection via the tables
// V_i[k] = c_i sum_b ec[c_i][b] T_r[x_i][b][k],  T_r[x][b][k] = sum_a er[r][a] Ep[x][a,b] D[a,b][k]
// (reference :327 in sample space).  One workgroup per image row: D lives in LDS, the row's pixels
// are counting-sorted by level, then levels are processed 8 at a time: build T_r for the batch
// (8 x nC x K' doubles in LDS), then one thread per pixel does its nC*K' multiply-adds and stores
// its K' outputs.  p*(nC+... ) work per pixel becomes nC*K' instead of p*K'.
constexpr int kPhXB = 4;     // levels per batch
constexpr int kPhPB = 128;   // pixels per output sub-chunk (staged through LDS for 16-byte stores)
constexpr int kPhKmax = 64;  // eigenvectors handled by this kernel

template <int KT>  // Kp = 16 KT >= K: sT rows are zero padded to Kp so that the pixel loop has no per-k branch
__global__ __launch_bounds__(256) void k_project_hist(const float* __restrict__ lum, GridSpec gs, int p, int row0,
                                                      const double* __restrict__ er, const double* __restrict__ ecT,
                                                      const double* __restrict__ Ep, const double* __restrict__ Dm,
                                                      int ldd, int K, const double* __restrict__ cvec,
                                                      float* __restrict__ V, int ldv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nC = gs.nSelCols, nR = gs.nSelRows, W = gs.W;
    constexpr int Kp = 16 * KT;
    double* sD = reinterpret_cast<double*>(smem_raw);        // [p][K]
    double* sT = sD + (size_t)p * K;                         // [XB][nC][Kp]
    double* sE = sT + (size_t)kPhXB * nC * Kp;              // [XB][p]  er[r][a] * Ep[x][a,b]
    int* cnt = reinterpret_cast<int*>(sE + (size_t)kPhXB * p);  // [257] level offsets
    int* fill = cnt + kLevels + 4;                           // [256]  (256 + 4 + 256 ints keep sOut 16-byte aligned)
    float* sOut = reinterpret_cast<float*>(fill + kLevels);  // [PB][ldv]
    unsigned short* idx = reinterpret_cast<unsigned short*>(sOut + (size_t)kPhPB * ldv);  // [W]
    const int tid = threadIdx.x, lrow = blockIdx.x, r = row0 + lrow;
    const double* er_r = er + (size_t)lrow * nR;
    for (int i = tid; i < p * K; i += 256) sD[i] = Dm[(size_t)(i / K) * ldd + (i % K)];
    for (int i = tid; i < kLevels; i += 256) {
        cnt[i] = 0;
        fill[i] = 0;
    }
    if (tid == 0) cnt[kLevels] = 0;
    __syncthreads();
    const float* lrowp = lum + (size_t)r * W;
    for (int c = tid; c < W; c += 256) atomicAdd(&cnt[(int)lrowp[c]], 1);
    __syncthreads();
    if (tid < 64) {  // exclusive scan of 256 counters by one wave: 4 per lane + wave scan
        int v[4], s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = cnt[4 * tid + j];
            s += v[j];
        }
        int incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (tid >= off) incl += t;
        }
        int base = incl - s;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cnt[4 * tid + j] = base;
            base += v[j];
        }
        if (tid == 63) cnt[kLevels] = base;
    }
    __syncthreads();
    for (int c = tid; c < W; c += 256) {
        const int x = (int)lrowp[c];
        idx[cnt[x] + atomicAdd(&fill[x], 1)] = (unsigned short)c;
    }
    __syncthreads();
    for (int x0 = 0; x0 < kLevels; x0 += kPhXB) {
        const int j0 = cnt[x0], j1 = cnt[x0 + kPhXB];
        if (j1 == j0) continue;  // uniform: no pixel of this row has one of these levels
        // stage er[r][a] * Ep[x][a,b] for the batch (coalesced), then T[xl][b][k] = sum_a sE[xl][a,b] D[a,b][k]
        for (int i = tid; i < kPhXB * p; i += 256) {
            const int sidx = i % p;
            sE[i] = er_r[sidx / nC] * Ep[(size_t)x0 * p + i];
        }
        __syncthreads();
        {
            constexpr int kc = Kp / 4;  // chunks of 4 consecutive k (zero beyond K)
            for (int o = tid; o < kPhXB * nC * kc; o += 256) {
                const int xl = o / (nC * kc), rem = o - xl * nC * kc, b = rem / kc, k0 = (rem - b * kc) * 4;
                const double* se = sE + (size_t)xl * p + b;
                double t[4] = {0.0, 0.0, 0.0, 0.0};
                if (k0 + 3 < K) {
                    for (int a = 0; a < nR; ++a) {
                        const double ev = se[a * nC];
                        const double* dp = sD + (size_t)(a * nC + b) * K + k0;
                        t[0] += ev * dp[0];
                        t[1] += ev * dp[1];
                        t[2] += ev * dp[2];
                        t[3] += ev * dp[3];
                    }
                } else if (k0 < K) {
                    for (int a = 0; a < nR; ++a) {
                        const double ev = se[a * nC];
                        const double* dp = sD + (size_t)(a * nC + b) * K + k0;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (k0 + q < K) t[q] += ev * dp[q];
                    }
                }
                double* tp = sT + ((size_t)xl * nC + b) * Kp + k0;
                tp[0] = t[0];
                tp[1] = t[1];
                tp[2] = t[2];
                tp[3] = t[3];
            }
        }
        __syncthreads();
        for (int jb = j0; jb < j1; jb += kPhPB) {
            const int npix = min(kPhPB, j1 - jb);
            if (tid < npix) {
                const int c = idx[jb + tid];
                const int xl = (int)lrowp[c] - x0;
                const double cf = cvec[(size_t)lrow * W + c];
                double acc[Kp];
#pragma unroll
                for (int k = 0; k < Kp; ++k) acc[k] = 0.0;
                const double* Tx = sT + (size_t)xl * nC * Kp;
                for (int b = 0; b < nC; ++b) {
                    const double e = ecT[(size_t)b * W + c];
                    const double* Tb = Tx + (size_t)b * Kp;
#pragma unroll
                    for (int k = 0; k < Kp; ++k) acc[k] += e * Tb[k];
                }
                float* so = sOut + (size_t)tid * ldv;
#pragma unroll
                for (int k = 0; k < Kp; ++k)
                    if (k < ldv) so[k] = (float)(cf * acc[k]);  // columns K..ldv-1 are exact zeros (T is zero padded)
            }
            __syncthreads();
            // whole rows of V, 16 bytes per lane
            const int nq = ldv >> 2;
            for (int it = tid; it < npix * nq; it += 256) {
                const int pj = it / nq, q = it - pj * nq;
                const int c = idx[jb + pj];
                *reinterpret_cast<float4*>(V + ((size_t)lrow * W + c) * ldv + 4 * q) =
                    *reinterpret_cast<const float4*>(sOut + (size_t)pj * ldv + 4 * q);
            }
            __syncthreads();
        }
    }
}

bool project_hist_ok(GridSpec gs, int p, int K) {
    if (K > kPhKmax || gs.W > 65535) return false;
    const size_t Kp = ((size_t)K + 15) & ~(size_t)15;
    const size_t shm = ((size_t)p * K + (size_t)kPhXB * gs.nSelCols * Kp + (size_t)kPhXB * p) * sizeof(double) +
                       (size_t)(2 * kLevels + 4) * sizeof(int) + (size_t)kPhPB * (((size_t)K + 3) & ~(size_t)3) * sizeof(float) +
                       (size_t)gs.W * sizeof(unsigned short) + 16;
    return K <= kPhKmax && shm <= 150 * 1024;
}

hipError_t project_hist(hipStream_t s, const float* d_lum, GridSpec gs, int p, int row0, int nrows_local,
                        const double* d_er, const double* d_ecT, const double* d_Ep, const double* d_D, int ldd, int K,
                        const double* d_c, float* d_V, int ldv) {
    if (nrows_local <= 0) return hipSuccess;
    const int KT = (K + 15) / 16;
    const size_t shm = ((size_t)p * K + (size_t)kPhXB * gs.nSelCols * 16 * KT + (size_t)kPhXB * p) * sizeof(double) +
                       (size_t)(2 * kLevels + 4) * sizeof(int) + (size_t)kPhPB * ldv * sizeof(float) +
                       (size_t)gs.W * sizeof(unsigned short) + 16;
    hipError_t e = hipSuccess;
#define NLE_PH_CASE(T)                                                                                              \
    case T:                                                                                                         \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_project_hist<T>),                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                             \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((k_project_hist<T>), dim3((unsigned)nrows_local), dim3(256), shm, s, d_lum, gs, p, row0, \
                           d_er, d_ecT, d_Ep, d_D, ldd, K, d_c, d_V, ldv);                                          \
        break;
    switch (KT) {
        NLE_PH_CASE(1)
        NLE_PH_CASE(2)
        NLE_PH_CASE(3)
        NLE_PH_CASE(4)
        default:
            return hipErrorInvalidValue;
    }
#undef NLE_PH_CASE
    return hipGetLastError();
}

}  // namespace nlek
