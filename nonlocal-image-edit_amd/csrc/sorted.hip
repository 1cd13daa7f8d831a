// Level-sorted image rows: the N-sized halves of the table formulation (fused.hip, "quantised-luminance fast path")
// without LDS atomics.
//
// The table kernels of round 1 (k_hist_pix, k_ghist_rows) keep a 256 x nC histogram per image row in LDS and add nC
// (Sinkhorn / apply) or nC(nC+1)/2 (Gram) terms per pixel with ds_add_f64: LDS-atomic bound (0.155 of the HBM roofline,
// 75 % LDS busy), slower on flat or two-level images (same-address serialisation) and not bitwise reproducible
// (the order of the atomic adds follows wave scheduling).  tools/micro/hist_pix_variants.hip measured the
// alternatives on the cfg4 shape: 64-bit fixed-point adds -20 % on noisy images and slower on structured ones,
// sub-histograms nothing -- and this form 2.2x.
//
// The levels of an image never change between the 2T + 2 passes of a train, so each image row is counting-sorted by
// level ONCE (k_sort_rows, stable) and cut into at most kSortedThreads chunks of equal size, each of ONE level; chunk j
// of the m chunks of a level takes the level's sorted pixels j, j + m, j + 2m, ... .  A pass then gives every thread one
// chunk: it keeps the level's table row g[x][0..nC) in registers, walks its pixels (the column factors come from one LDS
// table E[|c - c_b|] = exp(-(c - c_b)^2 / hx^2), bit-identical to the ecT table of k_hist_tables) and accumulates its
// nC (or nC(nC+1)/2) sums in registers.  The chunks of a level sit in consecutive threads and are combined by a fixed
// binary tree through LDS.  No atomics anywhere: results are bitwise reproducible, and a flat image costs what a noisy
// one does.
//
// Storage: the column indices of chunk k of a row sit CONTIGUOUSLY in slot k of the row (CHP = chunk length rounded up
// to 4 entries, 8-byte aligned), so a pass thread fetches four pixels' indices with one 8-byte load, one block of four
// ahead of their use; the chunk descriptor and the level's table row of the NEXT image row are requested while the
// current row is still being combined.  (The round-2 profile of the first form -- one 2-byte index load per pixel, each
// row starting with three dependent global loads -- showed the kernel waiting on memory latency for most of its time.)
//
// Reference arithmetic restated: the Sinkhorn row products / column sums of src/filter.cpp:238-245, the Gram
// Wab Wab^T of :296 and the reduce half of apply (:456), exactly as fused.hip derives them; only the order of the
// fp64 sums differs.
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

#define NLE_PIXEL_FENCE() __builtin_amdgcn_sched_barrier(0)

namespace nlek {

namespace {
constexpr int kLevels = 256;
constexpr int kT = kSortedThreads;
constexpr int kSortThreads = 256;

// inplaceReciprocal, src/filter.cpp:42-54.  v_rcp_f64 + two Newton steps (|s| >= eps = 1e-10 keeps every step in range):
// 1 ulp, 6 instructions instead of the 13 of the IEEE division sequence -- these kernels are instruction-issue bound
__device__ __forceinline__ double recip0_d(double s, double eps) {
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    r = fma(fma(-s, r, 1.0), r, r);
    return (fabs(s) >= eps) ? r : 0.0;
}

// E[|c - c_b|] from the LDS table, given the pre-scaled 16-bit operands c8 = 8 c, cb8 = 8 c_b (8 W <= 65536) and the LDS
// byte address of the table: ONE v_sad_u16 (|c8 - cb8| + base) makes the address -- no subtract / negate / max / shift /
// base add -- and the value is read through an LDS-address-space pointer
using lds_cdouble_ptr = const __attribute__((address_space(3))) double*;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ double e_at(unsigned sE_addr, unsigned c8, unsigned cb8) {
    const unsigned a = __builtin_amdgcn_sad_u16(c8, cb8, sE_addr);
    return *(lds_cdouble_ptr)a;
}

// f(b, e_b) for b = 0 .. NC-1, e_b = exp(-(c - c_b)^2 / hx^2), c_b = cb0 + b cs: the column factors of one pixel.
//   REC == false: NC reads of the E table.  64 lanes read 64 unrelated addresses: ~3 lanes per bank on average, and at
//     NC reads per pixel the LDS pipe, not the VALU, bounds the pass kernels (round 2: 10 reads -> 84 us at cfg4).
//   REC == true: e_0 and e_1 from the table, the rest from the exact recurrence of a Gaussian on an equispaced grid,
//       e_{b+1} = e_b rho_b,   rho_{b+1} = rho_b kappa,   rho_0 = e_1 / e_0,   kappa = exp(-2 cs^2 / hx^2)
//     (2 reads + 2 (NC - 2) multiplies + one reciprocal).  Rounding: e_b carries O(b^2 / 2) ulp (NC = 10: ~5e-15
//     relative) -- the probes of profiles/r2_readme_pair_sensitivity.txt put 1e-13 affinity noise at 1e-11 on the layers.
//     The host enables it only where no e_b, rho_b leaves the normal range (sorted_recurrence).
template <int NC, bool REC, class F>
__device__ __forceinline__ void column_factors(unsigned sEa, unsigned c8, int cb0, int cs, double kappa, F&& f) {
    if constexpr (!REC || NC <= 2) {
#pragma unroll
        for (int b = 0; b < NC; ++b) f(b, e_at(sEa, c8, (unsigned)(cb0 + b * cs) << 3));
    } else {
        const double e0 = e_at(sEa, c8, (unsigned)cb0 << 3);
        double eb = e_at(sEa, c8, (unsigned)(cb0 + cs) << 3);
        f(0, e0);
        f(1, eb);
        double r = __builtin_amdgcn_rcp(e0);  // 1 / e_0 to 1 ulp: two Newton steps (e_0 is a normal number here)
        r = fma(fma(-e0, r, 1.0), r, r);
        r = fma(fma(-e0, r, 1.0), r, r);
        double rho = eb * r;
#pragma unroll
        for (int b = 2; b < NC; ++b) {
            rho *= kappa;
            eb *= rho;
            f(b, eb);
        }
    }
}

// block-wide sum / max of one int per thread (256 threads), result in every thread; `red`: 8 ints of LDS
__device__ __forceinline__ int block_sum256(int v, int* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ int block_max256(int v, int* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return max(max(red[0], red[1]), max(red[2], red[3]));
}
}  // namespace

int sorted_max_width() { return 8192; }  // 8 W fits 16 bits; chunks of <= 32 pixels (kMaxBlocks)

// E[d] = exp(-d^2 / hx^2), d = 0 .. W: the same expression as ecT in k_hist_tables (fused.hip), so the two agree bit for bit
__global__ void k_dist_table(int W, double inv_hx2, double* __restrict__ E) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d <= W) {
        const double dd = (double)d;
        E[d] = exp(-(dd * dd) * inv_hx2);
    }
}

hipError_t dist_table(hipStream_t s, int W, double hx, double* d_E) {
    hipLaunchKernelGGL(k_dist_table, dim3((unsigned)((W + 256) / 256)), dim3(256), 0, s, W, 1.0 / (hx * hx), d_E);
    return hipGetLastError();
}

// ------------------------------------------------------------------ once per train: sort every row by level
// Chunk length bound: CH is the smallest length with sum_x ceil(tot[x] / CH) <= kT; with at most 256 non-empty levels
// that sum is <= wn / CH + 256, so CH <= ceil(W / 256) and a slot never needs more than sorted_chp_max(W) entries.
__host__ __device__ inline int sorted_chp_max(int W) { return (((W + 255) / 256) + 3) & ~3; }
constexpr int kMaxBlocks = 8;  // blocks of four indices per chunk: sorted_chp_max(sorted_max_width()) / 4
// entries per row of the scol buffer: kT slots of the largest size + the 4 entries a thread reads ahead of its slot
__host__ __device__ inline size_t sorted_row_pitch(int W) { return (size_t)kT * sorted_chp_max(W) + 4; }

// chunk descriptor: x = len | level << 6 | j << 14 | steps << 23,  y = m | CHP << 16
//   len    pixels of the chunk (0: idle thread),  j, m: the chunk is number j of the m chunks of its level,
//   steps  ceil(log2(max m of the row)) = depth of the combine tree,  CHP: slot size of this row (entries)
__device__ __forceinline__ int dsc_len(uint2 d) { return (int)(d.x & 63u); }
__device__ __forceinline__ int dsc_level(uint2 d) { return (int)((d.x >> 6) & 255u); }
__device__ __forceinline__ int dsc_j(uint2 d) { return (int)((d.x >> 14) & 511u); }
__device__ __forceinline__ int dsc_steps(uint2 d) { return (int)((d.x >> 23) & 15u); }
__device__ __forceinline__ int dsc_m(uint2 d) { return (int)(d.y & 0xffffu); }
__device__ __forceinline__ int dsc_chp(uint2 d) { return (int)(d.y >> 16); }

// One workgroup (256 threads) per local image row.  Sample pixels are left out (the N-sized sums skip them: their Phi
// rows are the exact V_A rows, reference :275).  Outputs, per row:
//   scol[pitch]  slot k (CHP entries, zero padded) = 8 x column (the byte offset the pass kernels feed to v_sad_u16) of
//                the pixels of chunk k in column order; chunk j of a level's m chunks holds the level's pixels
//                j, j + m, j + 2m, ... of a STABLE counting sort, so that the summation order of every later pass is a
//                function of the image alone;
//   desc[kT]     one chunk per pass thread (see above);
//   first[258]   first[x] = first chunk of level x (x = 0..256), first[257] = number of tree steps.
__global__ __launch_bounds__(kSortThreads) void k_sort_rows(const float* __restrict__ lum, GridSpec gs, int row0,
                                                            unsigned short* __restrict__ scol, uint2* __restrict__ desc,
                                                            unsigned short* __restrict__ first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W = gs.W;
    unsigned short* srow = reinterpret_cast<unsigned short*>(smem_raw);  // [kT * CHP + 4] the row's slots
    __shared__ int tot[kLevels], off[kLevels + 1], run[kLevels], fch[kLevels + 1], red[8];
    __shared__ float rcpm[kLevels];
    __shared__ __attribute__((aligned(16))) unsigned char cntG[4][kLevels];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = blockIdx.x, r = row0 + lrow;
    const float* lrow_p = lum + (size_t)r * W;
    const int dr = r - gs.rowOff;
    const bool sample_row = dr >= 0 && (dr % gs.rowStep) == 0 && (dr / gs.rowStep) < gs.nSelRows;
    auto is_sample = [&](int c) {
        if (!sample_row) return false;
        const int dc = c - gs.colOff;
        return dc >= 0 && (dc % gs.colStep) == 0 && (dc / gs.colStep) < gs.nSelCols;
    };
    tot[tid] = 0;
    run[tid] = 0;
    __syncthreads();
    for (int c = tid; c < W; c += kSortThreads)
        if (!is_sample(c)) atomicAdd(&tot[(int)lrow_p[c]], 1);  // integer counts: order does not matter
    __syncthreads();
    // exclusive prefix over the 256 levels (Hillis-Steele on `off`)
    const int mine = tot[tid];
    off[tid + 1] = mine;
    if (tid == 0) off[0] = 0;
    __syncthreads();
    for (int d = 1; d < kLevels; d <<= 1) {
        const int v = (tid + 1 > d) ? off[tid + 1 - d] : 0;
        __syncthreads();
        if (tid + 1 > d) off[tid + 1] += v;
        __syncthreads();
    }
    const int wn = off[kLevels];  // non-sample pixels of the row
    // smallest chunk size CH with sum_x ceil(tot[x] / CH) <= kT (monotone in CH): every thread of a pass gets at most
    // one chunk, and no chunk is longer than CH
    int lo = max(1, (wn + kT - 1) / kT), hi = max(1, block_max256(mine, red));
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int n = block_sum256((mine + mid - 1) / mid, red);
        if (n <= kT) hi = mid;
        else lo = mid + 1;
    }
    const int CH = lo, CHP = (CH + 3) & ~3;
    const int m = (mine + CH - 1) / CH;
    const int maxm = block_max256(m, red);
    rcpm[tid] = m > 0 ? 1.0f / (float)m : 0.f;
    fch[tid + 1] = m;
    if (tid == 0) fch[0] = 0;
    __syncthreads();
    for (int d = 1; d < kLevels; d <<= 1) {
        const int v = (tid + 1 > d) ? fch[tid + 1 - d] : 0;
        __syncthreads();
        if (tid + 1 > d) fch[tid + 1] += v;
        __syncthreads();
    }
    int steps = 0;
    while ((1 << steps) < maxm) ++steps;
    unsigned short* frow = first + (size_t)lrow * 258;
    frow[tid] = (unsigned short)fch[tid];
    if (tid == 0) {
        frow[kLevels] = (unsigned short)fch[kLevels];
        frow[kLevels + 1] = (unsigned short)steps;
    }
    const int nchunks = fch[kLevels];
    for (int k = tid; k < kT; k += kSortThreads) {
        uint2 d = make_uint2((unsigned)steps << 23, 1u | ((unsigned)CHP << 16));  // idle: len 0, m 1
        if (k < nchunks) {
            int a = 0, b = kLevels;  // largest x with fch[x] <= k
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (fch[mid] <= k) a = mid;
                else b = mid;
            }
            const int x = a, j = k - fch[x], mx = fch[x + 1] - fch[x], cnt = tot[x];
            const int len = (cnt - j + mx - 1) / mx;
            d = make_uint2((unsigned)len | ((unsigned)x << 6) | ((unsigned)j << 14) | ((unsigned)steps << 23),
                           (unsigned)mx | ((unsigned)CHP << 16));
        }
        desc[(size_t)lrow * kT + k] = d;
    }
    const int used = nchunks * CHP + 4;  // entries a pass can read: the slots and one block past the last
    for (int i = tid; i < used; i += kSortThreads) srow[i] = 0;  // padding must be a valid column
    __syncthreads();
    // stable placement, four 64-column tiles (one per wave) at a time
    for (int c0 = 0; c0 < W; c0 += 4 * 64) {
        const int c = c0 + wave * 64 + lane;
        const bool valid = c < W && !is_sample(c);
        const int x = valid ? (int)lrow_p[c] : 0;
        // lanes of my level: AND over the 8 bits of x of (bit ? ballot(bit) : ~ballot(bit)), among the valid lanes
        unsigned long long eq = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long bm = __ballot((x >> bit) & 1);
            eq &= ((x >> bit) & 1) ? bm : ~bm;
        }
        const int rank = __popcll(eq & ((1ull << lane) - 1ull));
        const int count = __popcll(eq);
        reinterpret_cast<unsigned int*>(&cntG[0][0])[tid] = 0u;  // 256 threads x 4 B = the whole 4 x 256 B table
        __syncthreads();
        if (valid && rank == 0) cntG[wave][x] = (unsigned char)count;  // one writer per (tile, level); count <= 64
        __syncthreads();
        if (valid) {
            int prev = 0;
            for (int w2 = 0; w2 < wave; ++w2) prev += cntG[w2][x];
            // q-th pixel of level x (sorted order) -> chunk q mod m, position q div m; the quotient through a float
            // reciprocal is exact here: (q + 0.5) / m lies >= 0.5 / 512 from an integer, its value is <= CH + 1 <= 64
            const int q = run[x] + prev + rank, mx = fch[x + 1] - fch[x];
            const int pos = (int)(((float)q + 0.5f) * rcpm[x]);
            const int j = q - pos * mx;
            srow[(fch[x] + j) * CHP + pos] = (unsigned short)(c << 3);  // pre-scaled: 8 c (W <= 8192)
        }
        __syncthreads();
        run[tid] += cntG[0][tid] + cntG[1][tid] + cntG[2][tid] + cntG[3][tid];
        __syncthreads();
    }
    unsigned int* out = reinterpret_cast<unsigned int*>(scol + (size_t)lrow * sorted_row_pitch(W));
    const unsigned int* src = reinterpret_cast<const unsigned int*>(srow);
    for (int i = tid; i < used / 2; i += kSortThreads) out[i] = src[i];
}

// elements of the scol buffer
size_t sorted_scol_elems(int W, int nrows_local) { return (size_t)std::max(nrows_local, 0) * sorted_row_pitch(W) + 16; }

hipError_t sort_rows(hipStream_t s, const float* d_lum, GridSpec gs, int row0, int nrows_local, unsigned short* d_scol,
                     uint2* d_desc, unsigned short* d_first) {
    if (gs.W > sorted_max_width() || nrows_local <= 0) return nrows_local <= 0 ? hipSuccess : hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)nrows_local), dim3(kSortThreads),
                       sorted_row_pitch(gs.W) * sizeof(unsigned short), s, d_lum, gs, row0, d_scol, d_desc, d_first);
    return hipGetLastError();
}

// ------------------------------------------------------------------ LDS layout shared by the two pass kernels
// sE [W + 1] doubles | sP [kT][PS] doubles | sfirst [2][260] u16 (this row's and the next row's) | sCk [40] doubles (the
// quadratic factors of the moment form of k_sorted_pass)
__host__ __device__ inline size_t sorted_lds_bytes(int W, int ps) {
    return ((size_t)((W + 2) & ~1) + (size_t)kT * ps + 40) * sizeof(double) + 2 * 260 * sizeof(unsigned short);
}

// Combines the per-chunk partial sums v[0..NV) of the threads of one level (consecutive threads, j = position in the
// level's segment of m chunks) with a fixed binary tree through sP, `steps` = ceil(log2(max m of the row)).  On return
// sP[first_chunk_of_level][0..NV) holds the level's sums.  All threads of the workgroup must call it.
template <int NV, int PS>
__device__ __forceinline__ void combine_chunks(double (&v)[NV], double* sP, int tid, bool active, int j, int m, int steps) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sP[tid * PS + i] = v[i];
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int d = 1 << s;
        const bool part = active && (j & (2 * d - 1)) == 0 && j + d < m;
        if (part) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += sP[(tid + d) * PS + i];
        }
        __syncthreads();
        if (part) {
#pragma unroll
            for (int i = 0; i < NV; ++i) sP[tid * PS + i] = v[i];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ Sinkhorn half-iteration / apply reduce half
// For every local image row r (persistent workgroups, rows r = blockIdx.x, + gridDim.x, ...):
//   y_i = 1 (COLSUM) | recip(sum_b ec[c_i][b] g_r[x_i][b]) (RECIP) | c_i x_i (XVEC),   h_r[x][b] = sum_{i: x_i = x} ec[c_i][b] y_i
// g, hout: [nrows][b][x] (b-major tables, as k_hist_g writes and k_hist_hh reads them).
//
// CF: how a pixel gets its nC column factors e_b = exp(-(c - c_b)^2 / hx^2).
//   0  nC reads of the LDS table E (64 lanes, 64 unrelated addresses: ~8 cycles of the CU's one LDS pipe per read -- at
//      nC = 10 that pipe, not the vector units, bounds the pixel loop: 64 wave-pixels x 10 reads x 8 cycles = 2.1 us per row)
//   1  e_0, e_1 from the table, the rest by the recurrence of column_factors (nC > 12)
//   2  MOMENTS: on the equispaced grid e_b = e_0 rho^b C_b with rho = e_1 / e_0 (per pixel) and C_b = kappa^(b (b - 1) / 2)
//      (per grid), so the row product is a POLYNOMIAL in rho, sum_b e_b g_b = e_0 sum_b (C_b g_b) rho^b -- Horner on
//      coefficients the thread keeps for its chunk -- and the per-level sums are MOMENTS, sum_i e_b y_i = C_b sum_i (y_i e_0)
//      rho^b, scaled by C_b once per chunk: two table reads and ~3.5 nC fp64 operations per pixel, no e_b ever formed.
//      Exact algebra; the rounding is the recurrence's (a product of b factors).  The host enables it where no power
//      leaves the normal range (sorted_moments_ok).
template <int NC, int CF>
__global__ __launch_bounds__(kT, (NC <= 12 ? 4 : 2)) void k_sorted_pass(int mode, const unsigned short* __restrict__ scol,
                                                    const uint2* __restrict__ desc, const unsigned short* __restrict__ first,
                                                    GridSpec gs, int row0, int nrows, const double* __restrict__ Etab,
                                                    const double* __restrict__ g, double eps, double* __restrict__ ybuf,
                                                    double* __restrict__ hout, const double* __restrict__ cvec,
                                                    const float* __restrict__ xvec, double kappa, int lev_t0, int lev_nt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int n = kLevels * NC;
    constexpr int SL = NC < 11 ? NC : 11;  // sums combined per tree (slices of the nC sums when nC > 11)
    constexpr int PS = SL | 1;             // odd stride: consecutive threads' rows start on different banks
    constexpr bool KEEP_E = NC <= 32;      // keep the column factors of a pixel in registers between the two loops (254 VGPRs at NC = 30, no spills)
    constexpr bool REC = CF == 1, MOM = CF == 2;
    const int W = gs.W;
    const size_t pitch = sorted_row_pitch(W);
    double* sE = reinterpret_cast<double*>(smem_raw);
    double* sP = sE + ((W + 2) & ~1);
    unsigned short* sfirst = reinterpret_cast<unsigned short*>(sP + (size_t)kT * PS);  // [2][260], rows alternate
    double* sCk = reinterpret_cast<double*>(sfirst + 2 * 260);                          // [NC] C_b = kappa^(b (b - 1) / 2)
    const int tid = threadIdx.x;
    for (int i = tid; i <= W; i += kT) sE[i] = Etab[i];
    if (MOM && tid == 0) {
        double ck = 1.0, kp = 1.0;  // C_{b+1} = C_b kappa^b
        for (int b = 0; b < NC; ++b) {
            sCk[b] = ck;
            ck *= kp;
            kp *= kappa;
        }
    }
    const int cb0 = gs.colOff, cs = gs.colStep;
    const bool recip = mode == ROWPASS_RECIP, xmode = mode == ROWPASS_XVEC;
    const unsigned sEa = lds_addr(sE);

    // Software pipeline over the rows of this workgroup: everything a row needs from global memory is requested while
    // the PREVIOUS row is being combined -- its first indices and its level's table row (their address needs the row's
    // chunk descriptor, which is therefore requested two rows ahead) -- and taken in before that row's table is stored.
    // Nothing foreign is pending during the pixel loop, so its waits (the counter retires in order) are exact.
    const int G = (int)gridDim.x;  // <= nrows
    int lrow = blockIdx.x, nrow = lrow + G;
    uint2 dsc = desc[(size_t)lrow * kT + tid];
    uint2 dsc_n = dsc;
    if (nrow < nrows) dsc_n = desc[(size_t)nrow * kT + tid];
    if (tid < 258) sfirst[tid] = first[(size_t)lrow * 258 + tid];
    // all indices of the chunk (<= kMaxBlocks blocks of four) live in registers: the pixel loop itself loads nothing
    uint2 idx[kMaxBlocks];
    auto load_idx = [&](uint2 (&dst)[kMaxBlocks], const int row, const uint2 d) {
        const uint2* slot = reinterpret_cast<const uint2*>(scol + (size_t)row * pitch + (size_t)tid * dsc_chp(d));
        const int ln = dsc_len(d);
#pragma unroll
        for (int b = 0; b < kMaxBlocks; ++b) {  // blocks past the chunk: column 0 (the wave may walk further than this lane)
            uint2 v = make_uint2(0u, 0u);
            if (4 * b < ln) v = slot[b];
            dst[b] = v;
        }
    };
    load_idx(idx, lrow, dsc);
    double gv[NC];
#pragma unroll
    for (int b = 0; b < NC; ++b) gv[b] = recip ? g[(size_t)lrow * n + b * kLevels + dsc_level(dsc)] : 0.0;
    __syncthreads();  // sE, sfirst visible
    for (int par = 0;; par ^= 1) {
        const bool has_next = nrow < nrows;
        const int nnrow = nrow + G;
        const unsigned short* sfc = sfirst + par * 260;
        const int len = dsc_len(dsc);
        double acc[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) acc[b] = 0.0;
        const double* cv_row = cvec ? cvec + (size_t)lrow * W : nullptr;
        const float* xv_row = xvec ? xvec + (size_t)(row0 + lrow) * W : nullptr;
        double* yb_row = ybuf ? ybuf + (size_t)lrow * W : nullptr;
        // requested BEFORE the pixel loop, in flight under it: the descriptor of the row after next, the next row's
        // first-chunk table entry and its indices (the pixel loop has no wait on the memory counter)
        uint2 dsc_nn = dsc_n;
        unsigned short sf_n = 0;
        uint2 idx_n[kMaxBlocks];
        if (has_next) {
            if (nnrow < nrows) dsc_nn = desc[(size_t)nnrow * kT + tid];
            sf_n = first[(size_t)nrow * 258 + (tid < 258 ? tid : 257)];
            load_idx(idx_n, nrow, dsc_n);
        }
        // MOM: the next row's table row too.  With it requested only after the loop, a workgroup's row is loop (2 us of
        // arithmetic) + a memory round trip + combine + stores (5-7 us of waiting): two workgroups per CU keep the vector
        // units busy for less than half of that.  The moment form leaves the registers for it (no column factor is kept).
        constexpr bool PRE = MOM && (NC <= 11 || (NC > 12 && NC <= 24));  // where the registers allow it without spilling
        double gvn[PRE ? NC : 1];
        if constexpr (PRE) {
            if (has_next && recip) {
#pragma unroll
                for (int b = 0; b < NC; ++b) gvn[b] = g[(size_t)nrow * n + b * kLevels + dsc_level(dsc_n)];
            }
        }
        // One pixel at a time (the column factors of one pixel fill the registers; the scheduling barrier keeps the
        // unrolled bodies from being interleaved), up to the longest chunk of the WAVE: a scalar bound, so that a wave
        // whose chunks are all short skips the rest without per-lane bookkeeping.  Lanes past their own chunk add zeros.
        double(&q)[NC] = gv;  // MOM: the polynomial's coefficients C_b g_b of this thread's chunk, in place of g_b
        if constexpr (MOM) {
            if (recip) {
#pragma unroll
                for (int b = 2; b < NC; ++b) gv[b] *= sCk[b];  // C_0 = C_1 = 1
            }
        }
        // MOM: two (nC <= 12) or FOUR pixels of the chunk at a time.  A pixel's work is a few dependent chains of fp64
        // operations -- the reciprocal's Newton steps, the Horner recurrences, the running powers -- and with 2-4 waves per
        // SIMD their LATENCY, not their issue rate, is what the loop costs; four independent pixels give the scheduler
        // 8-16 chains to interleave, and the state of a pixel is 8 doubles now that no e_b is kept.  Lanes past their own
        // chunk (and the padding of the last block) work on column 0 and add exact zeros.
        constexpr int PB = NC > 0 ? 1 : 2;  // pixels in flight per thread: interleaving 2 / 4 bought nothing (profiles/r4_pass_ablation.txt);
                               // one leaves the registers for the next row's table row, requested BEFORE the loop (below)
        auto pixels = [&](const unsigned (&c8)[PB], const int base) {
            if constexpr (!MOM) return;
            else {
            double e0[PB], rho[PB], rho2[PB], y[PB];
            bool keep[PB];
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                e0[k] = e_at(sEa, c8[k], (unsigned)cb0 << 3);
                rho[k] = e_at(sEa, c8[k], (unsigned)(cb0 + cs) << 3);
                keep[k] = base + k < len;
                y[k] = 1.0;
            }
            if (xmode) {
#pragma unroll
                for (int k = 0; k < PB; ++k) y[k] = cv_row[c8[k] >> 3] * (double)xv_row[c8[k] >> 3];  // apply: y_i = c_i x_i
            }
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                double r0 = __builtin_amdgcn_rcp(e0[k]);  // e_0 is a normal number here (sorted_moments_ok)
                r0 = fma(fma(-e0[k], r0, 1.0), r0, r0);
                r0 = fma(fma(-e0[k], r0, 1.0), r0, r0);
                rho[k] *= r0;
                rho2[k] = rho[k] * rho[k];
            }
            if (recip) {
                // even and odd coefficients: two Horner chains in rho^2 of half the length, per pixel
                constexpr int LE = (NC - 1) & ~1, LO = ((NC - 2) & ~1) + 1;   // highest even / odd index < NC
                double se[PB], so[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    se[k] = q[LE];
                    so[k] = q[LO];
                }
#pragma unroll
                for (int b = LE - 2; b >= 0; b -= 2) {
#pragma unroll
                    for (int k = 0; k < PB; ++k) se[k] = fma(se[k], rho2[k], q[b]);
                }
#pragma unroll
                for (int b = LO - 2; b >= 1; b -= 2) {
#pragma unroll
                    for (int k = 0; k < PB; ++k) so[k] = fma(so[k], rho2[k], q[b]);
                }
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const double sm = fma(so[k], rho[k], se[k]) * e0[k];
                    double r = __builtin_amdgcn_rcp(sm);  // inplaceReciprocal (src/filter.cpp:42-54), recip0_d's arithmetic
                    r = fma(fma(-sm, r, 1.0), r, r);
                    r = fma(fma(-sm, r, 1.0), r, r);
                    y[k] = r;
                    keep[k] = keep[k] && fabs(sm) >= eps;
                }
            }
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const bool on = base + k < len;
                y[k] = keep[k] ? y[k] : 0.0;  // the padding of the slot adds exact zeros
                if (yb_row != nullptr && on) yb_row[c8[k] >> 3] = y[k];
            }
            // moments: acc_b += sum_k y_k e_0k rho_k^b, even and odd chains per pixel, the pixels summed pairwise
            auto sum_pb = [](const double (&t)[PB]) {
                if constexpr (PB == 4) return (t[0] + t[1]) + (t[2] + t[3]);
                else if constexpr (PB == 2) return t[0] + t[1];
                else return t[0];
            };
            double te[PB], to[PB];
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                te[k] = y[k] * e0[k];
                to[k] = te[k] * rho[k];
            }
            acc[0] += sum_pb(te);
            acc[1] += sum_pb(to);
#pragma unroll
            for (int b = 2; b < NC; b += 2) {
#pragma unroll
                for (int k = 0; k < PB; ++k) te[k] *= rho2[k];
                acc[b] += sum_pb(te);
                if (b + 1 < NC) {
#pragma unroll
                    for (int k = 0; k < PB; ++k) to[k] *= rho2[k];
                    acc[b + 1] += sum_pb(to);
                }
            }
            }
        };
        auto pixel = [&](const unsigned c8, const bool on) {
            double e[KEEP_E ? NC : 1];
            double y = 1.0;
            bool keep = on;
            if (recip) {
                double s0 = 0.0, s1 = 0.0;
                column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) {
                    if constexpr (KEEP_E) e[b] = ev;
                    if (b & 1) s1 += ev * gv[b];
                    else s0 += ev * gv[b];
                });
                // inplaceReciprocal (src/filter.cpp:42-54): 1 / s, 0 where |s| < eps; recip0_d's arithmetic
                const double sm = s0 + s1;
                double r = __builtin_amdgcn_rcp(sm);
                r = fma(fma(-sm, r, 1.0), r, r);
                r = fma(fma(-sm, r, 1.0), r, r);
                y = r;
                keep = on && fabs(sm) >= eps;
            } else {
                if (xmode) y = cv_row[c8 >> 3] * (double)xv_row[c8 >> 3];  // apply: y_i = c_i x_i
                if constexpr (KEEP_E) column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) { e[b] = ev; });
            }
            y = keep ? y : 0.0;  // the padding of the slot adds exact zeros
            if (yb_row != nullptr && on) yb_row[c8 >> 3] = y;
            if constexpr (KEEP_E) {
#pragma unroll
                for (int b = 0; b < NC; ++b) acc[b] += e[b] * y;
            } else {
                column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) { acc[b] += ev * y; });
            }
        };
        int wlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wlen = max(wlen, __shfl_xor(wlen, off));
        wlen = __builtin_amdgcn_readfirstlane(wlen);
        if constexpr (MOM) {
#pragma unroll
            for (int b = 0; b < kMaxBlocks; ++b) {
                if (4 * b >= wlen) break;
                if constexpr (PB == 4) {
                    const unsigned c8[PB] = {idx[b].x & 0xffffu, idx[b].x >> 16, idx[b].y & 0xffffu, idx[b].y >> 16};
                    pixels(c8, 4 * b);
                    NLE_PIXEL_FENCE();
                } else if constexpr (PB == 2) {
                    const unsigned c8a[PB] = {idx[b].x & 0xffffu, idx[b].x >> 16}, c8b[PB] = {idx[b].y & 0xffffu, idx[b].y >> 16};
                    pixels(c8a, 4 * b);
                    NLE_PIXEL_FENCE();
                    if (4 * b + 2 < wlen) {
                        pixels(c8b, 4 * b + 2);
                        NLE_PIXEL_FENCE();
                    }
                } else {
                    const unsigned cs4[4] = {idx[b].x & 0xffffu, idx[b].x >> 16, idx[b].y & 0xffffu, idx[b].y >> 16};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (4 * b + k < wlen) {
                            const unsigned c8one[PB] = {cs4[k]};
                            pixels(c8one, 4 * b + k);
                            NLE_PIXEL_FENCE();
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int b = 0; b < kMaxBlocks; ++b) {
            if (4 * b >= wlen) break;
            pixel(idx[b].x & 0xffffu, 4 * b < len);
            NLE_PIXEL_FENCE();
            if (4 * b + 1 < wlen) {
                pixel(idx[b].x >> 16, 4 * b + 1 < len);
                NLE_PIXEL_FENCE();
            }
            if (4 * b + 2 < wlen) {
                pixel(idx[b].y & 0xffffu, 4 * b + 2 < len);
                NLE_PIXEL_FENCE();
            }
            if (4 * b + 3 < wlen) {
                pixel(idx[b].y >> 16, 4 * b + 3 < len);
                NLE_PIXEL_FENCE();
            }
        }
        }
        // The combine / load / store phase is a chain of barriers and memory round trips; the other workgroup of the CU is
        // (usually) in its pixel loop and, being older or younger, wins or loses every issue slot wholesale (the stamps of
        // tools/stamps_run.sh: the second workgroup's combine took 4-5 us against 2.2 us alone).  Raise the priority here,
        // drop it for the loop.
        if constexpr (MOM) {
#pragma unroll
            for (int b = 2; b < NC; ++b) acc[b] *= sCk[b];  // C_0 = C_1 = 1
        }
        __builtin_amdgcn_s_setprio(3);
        // the next row's table row: in flight under the combine.  (Requested before the pixel loop -- registers
        // permitting -- the kernel gets SLOWER, as it does with an L2 prefetch: profiles/r2_pass_ablation.txt.)
        if (has_next && recip) {
            if constexpr (PRE) {
#pragma unroll
                for (int b = 0; b < NC; ++b) gv[b] = gvn[b];
            } else {
#pragma unroll
                for (int b = 0; b < NC; ++b) gv[b] = g[(size_t)nrow * n + b * kLevels + dsc_level(dsc_n)];
            }
        }
        double* hrow = hout + (size_t)lrow * n;
#pragma unroll
        for (int s0 = 0; s0 < NC; s0 += SL) {
            double v[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i) v[i] = (s0 + i < NC) ? acc[s0 + i] : 0.0;
            // The partial sums of the chunks of a level sit in consecutive threads.  They are written to LDS once and
            // every table entry (level x, column b) is then summed by ONE thread over the level's chunks in chunk order --
            // a fixed order, so still bitwise reproducible -- and stored straight away: two barriers per slice, where the
            // binary tree through LDS took 2 log2(max chunks per level) + 2 (ablation, profiles/r4_pass_ablation.txt:
            // the tree was 25 of 81 us at cfg4, 190 of 697 at cfg5).  A level of a flat row has up to 512 chunks: its nC
            // entries are then summed by nC threads, 512 LDS reads each (~2 us), about what the tree cost on such a row.
#pragma unroll
            for (int i = 0; i < SL; ++i) sP[tid * PS + i] = v[i];
            __syncthreads();
            if (s0 == 0) {
                // Were the loads above still pending behind the table stores below, the next row's first pixel would
                // wait for those stores to be acknowledged.  Take them in here (all lanes: a divergent use would leave
                // a load pending for the others).
                asm volatile("" ::"v"(sf_n), "v"(dsc_nn.x), "v"(dsc_nn.y));
#pragma unroll
                for (int b = 0; b < kMaxBlocks; ++b) asm volatile("" ::"v"(idx_n[b].x), "v"(idx_n[b].y));
#pragma unroll
                for (int b = 0; b < NC; ++b) asm volatile("" ::"v"(gv[b]));
                // next row's first-chunk table (its last readers left before the previous row's end barrier; its next
                // readers come after the barriers of the next combine)
                if (has_next && tid < 258) sfirst[(par ^ 1) * 260 + tid] = sf_n;
            }
            const int ns = (NC - s0 < SL) ? NC - s0 : SL;
            // only the level tiles that occur anywhere in the image are stored (k_hist_hh reads no others)
            const int nlev = lev_nt * 16, xlo = lev_t0 * 16;
            const float inv_nlev = 1.0f / (float)nlev;
            for (int i = tid; i < ns * nlev; i += kT) {
                const int bb = (int)(((float)i + 0.5f) * inv_nlev);  // i / nlev, exact: i < 12 * 256, (i + 0.5) / nlev is >= 1 / 512 off an integer
                const int xx = xlo + (i - bb * nlev);
                const int f0 = sfc[xx], f1 = sfc[xx + 1];
                double sum = 0.0;
                for (int t = f0; t < f1; ++t) sum += sP[t * PS + bb];
                hrow[(size_t)(s0 + bb) * kLevels + xx] = sum;
            }
            __syncthreads();  // before the next slice / row overwrites sP and sfirst
        }
        __builtin_amdgcn_s_setprio(0);  // (letting the two workgroups take turns at priority 1 in their loops balanced
                                        // their finish times, 65 / 71 us instead of 61 / 74, but not the kernel's)
        if (!has_next) break;
        lrow = nrow;
        nrow = nnrow;
        dsc = dsc_n;
        dsc_n = dsc_nn;
#pragma unroll
        for (int b = 0; b < kMaxBlocks; ++b) idx[b] = idx_n[b];
    }
}

static int sorted_grid(int nrows) {
    int ncu = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    int mult = 2;  // two 512-thread workgroups per CU, each walks its rows
    if (const char* e = std::getenv("NLE_SORTED_WGS_PER_CU")) mult = std::max(1, std::atoi(e));  // measurement hook
    return std::max(1, std::min(nrows, mult * ncu));
}

// Where the recurrence of column_factors stays inside the normal range of fp64 (with a wide margin): every e_b =
// exp(-(u - b cs)^2 / hx^2) and every ratio rho_b = exp((2 cs u - (2 b + 1) cs^2) / hx^2), u = c - cb0, c = 0 .. W-1.
// Narrow kernels (W / hx beyond ~20) keep the table form.  *kappa = exp(-2 cs^2 / hx^2).
bool sorted_recurrence(GridSpec gs, double hx, double* kappa) {
    const double cs = gs.colStep, umax = std::max<double>(gs.colOff, gs.W - 1 - gs.colOff), nC = gs.nSelCols;
    const double span = umax + nC * cs;  // |u - b cs| <= span
    const double m_e = span * span / (hx * hx);
    const double m_rho = (2.0 * cs * umax + (2.0 * nC + 1.0) * cs * cs) / (hx * hx);
    *kappa = std::exp(-2.0 * cs * cs / (hx * hx));
    const bool off = std::getenv("NLE_SORTED_TABLE") != nullptr;  // every column factor from the table
    // up to 12 columns the factors of a pixel stay in registers between its two uses (10 table reads per pixel at cfg4:
    // measured no slower than the recurrence, and bit-identical to the ecT table of the other kernels); beyond, the table
    // form reads every factor twice and the LDS pipe bounds it (cfg5, 30 columns: 1212 -> 861 us per pass)
    return !off && gs.nSelCols > 12 && m_e < 500.0 && m_rho < 500.0;
}

// Where the moment form of k_sorted_pass stays inside the normal range of fp64 with a wide margin: e_0 = exp(-u^2 / hx^2)
// (u = c - cb0), the powers rho^b = exp(b (2 cs u - cs^2) / hx^2), b < nC, and the running products y e_0 rho^b =
// y e_b / C_b <= y exp(nC^2 cs^2 / hx^2): every intermediate lies within exp(+-span^2 / hx^2), span = umax + nC cs, times the
// Sinkhorn scalings (<= 1 / eps = 1e10 each).  The bound 500 (1e217) leaves 1e90 on either side; cfg5's 30 columns at
// hx = W / 8 sit at 251.  Very narrow kernels (W / hx beyond ~20) keep the table / recurrence forms.
bool sorted_moments_ok(GridSpec gs, double hx) {
    if (std::getenv("NLE_SORTED_TABLE") != nullptr || std::getenv("NLE_SORTED_NO_MOMENTS") != nullptr) return false;
    const double cs = gs.colStep, umax = std::max<double>(gs.colOff, gs.W - 1 - gs.colOff), nC = gs.nSelCols;
    const double span = umax + nC * cs;
    return span * span / (hx * hx) < 500.0;
}

hipError_t sorted_pass(hipStream_t s, int mode, GridSpec gs, int row0, int nrows_local, const unsigned short* d_scol,
                       const uint2* d_desc, const unsigned short* d_first, const double* d_E, const double* d_g, double eps,
                       double* d_ybuf, double* d_h, const double* d_cvec, const float* d_xvec, bool rec, double kappa,
                       int lev_t0, int lev_nt, bool mom) {
    const int nC = gs.nSelCols;
    if (nC < 1 || nC > 36 || gs.W > sorted_max_width() || lev_t0 < 0 || lev_nt < 1 || lev_t0 + lev_nt > kLevels / 16)
        return hipErrorInvalidValue;
    if (nrows_local <= 0) return hipSuccess;
    const size_t shm = sorted_lds_bytes(gs.W, (nC < 11 ? nC : 11) | 1);
    const int grid = sorted_grid(nrows_local);
#define NLE_SP1(NCV, CFV)                                                                                                 \
    {                                                                                                                     \
        if (shm > 48 * 1024) {                                                                                            \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_pass<NCV, CFV>),                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                    \
            if (ea != hipSuccess) return ea;                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_sorted_pass<NCV, CFV>), dim3((unsigned)grid), dim3(kT), shm, s, mode, d_scol, d_desc,        \
                           d_first, gs, row0, nrows_local, d_E, d_g, eps, d_ybuf, d_h, d_cvec, d_xvec, kappa, lev_t0,     \
                           lev_nt);                                                                                       \
    }
#define NLE_SP(NCV)                                                                                                       \
    case NCV:                                                                                                             \
        if constexpr ((NCV) > 1) {                                                                                        \
            if (mom) {                                                                                                    \
                NLE_SP1(NCV, 2)                                                                                           \
                break;                                                                                                    \
            }                                                                                                             \
        }                                                                                                                 \
        if constexpr ((NCV) > 12) {                                                                                       \
            if (rec) NLE_SP1(NCV, 1) else NLE_SP1(NCV, 0)                                                                 \
        } else {                                                                                                          \
            if (rec) return hipErrorInvalidValue;                                                                         \
            NLE_SP1(NCV, 0)                                                                                               \
        }                                                                                                                 \
        break;
    switch (nC) {
        NLE_SP(1) NLE_SP(2) NLE_SP(3) NLE_SP(4) NLE_SP(5) NLE_SP(6) NLE_SP(7) NLE_SP(8) NLE_SP(9) NLE_SP(10) NLE_SP(11)
        NLE_SP(12) NLE_SP(13) NLE_SP(14) NLE_SP(15) NLE_SP(16) NLE_SP(17) NLE_SP(18) NLE_SP(19) NLE_SP(20)
        NLE_SP(21) NLE_SP(22) NLE_SP(23) NLE_SP(24) NLE_SP(25) NLE_SP(26) NLE_SP(27) NLE_SP(28) NLE_SP(29)
        NLE_SP(30) NLE_SP(31) NLE_SP(32) NLE_SP(33) NLE_SP(34) NLE_SP(35) NLE_SP(36)
        default: return hipErrorInvalidValue;
    }
#undef NLE_SP
#undef NLE_SP1
    return hipGetLastError();
}

// ------------------------------------------------------------------ apply, expand half, on the sorted rows
// out_l[i] = (float)(c_i sum_b ec[c_i][b] g_l[r][x_i][b]) for the nl <= 4 layers of a launch (g_l from w'_l = D (f_l o t),
// k_hist_g).  k_hist_dot does this pixel by pixel in image order with nl nC random LDS reads of the level-major tables per
// pixel (LDS-conflict bound: 376 us for four layers at cfg4 against a 160 us stream floor).  Here a thread keeps the nl
// table rows of ITS level in registers, so a pixel costs nC reads of the E table for all layers together; the results
// are scattered into the row's output buffer in LDS and leave it coalesced.  Sample pixels are not in the sorted lists:
// their slots hold whatever the buffer held and are overwritten by k_scatter_samples with the exact sample rows, as before.
// nC <= 12 (registers), W <= 4096 (LDS: E table + 4 rows of fp32 output).
// Wider grids (13 .. 36 columns) and wider images (W <= 8192): two layers per launch (2 nC table values in registers; LDS: the
// E table + two rows of output) and, where sorted_recurrence allows, the column factors by recurrence.
constexpr int kExpLayers = 4;
template <int NC, int kExpL, bool REC>
__global__ __launch_bounds__(kT) void k_sorted_expand(const unsigned short* __restrict__ scol, const uint2* __restrict__ desc,
                                                      GridSpec gs, int nrows, const double* __restrict__ Etab,
                                                      const double* __restrict__ g, size_t gstride, int nl,
                                                      const double* __restrict__ cvec, float* __restrict__ out,
                                                      long long ostride, double kappa, int round8) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int n = kLevels * NC;
    const int W = gs.W;
    double* sE = reinterpret_cast<double*>(smem_raw);
    float* sOut = reinterpret_cast<float*>(sE + ((W + 2) & ~1));  // [kExpL][W]
    const int tid = threadIdx.x;
    for (int i = tid; i <= W; i += kT) sE[i] = Etab[i];
    const int cb0 = gs.colOff, cs = gs.colStep;
    const unsigned sEa = lds_addr(sE);
    const size_t pitch = sorted_row_pitch(W);
    __syncthreads();
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        const uint2 dsc = desc[(size_t)lrow * kT + tid];
        const int len = dsc_len(dsc), x = dsc_level(dsc);
        const uint2* slot = reinterpret_cast<const uint2*>(scol + (size_t)lrow * pitch + (size_t)tid * dsc_chp(dsc));
        uint2 idx[kMaxBlocks];
#pragma unroll
        for (int b = 0; b < kMaxBlocks; ++b) {
            uint2 v = make_uint2(0u, 0u);
            if (4 * b < len) v = slot[b];
            idx[b] = v;
        }
        double gv[kExpL][NC];
#pragma unroll
        for (int l = 0; l < kExpL; ++l)
#pragma unroll
            for (int b = 0; b < NC; ++b) gv[l][b] = (l < nl) ? g[(size_t)l * gstride + (size_t)lrow * n + b * kLevels + x] : 0.0;
        const double* cv_row = cvec + (size_t)lrow * W;
        int wlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wlen = max(wlen, __shfl_xor(wlen, off));
        wlen = __builtin_amdgcn_readfirstlane(wlen);
        auto pixel = [&](const unsigned c8, const bool on) {
            const unsigned c = c8 >> 3;
            const double cv = cv_row[c];
            double e[NC];
            column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) { e[b] = ev; });
#pragma unroll
            for (int l = 0; l < kExpL; ++l) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int b = 0; b < NC; ++b) {
                    if (b & 1) s1 += e[b] * gv[l][b];
                    else s0 += e[b] * gv[l][b];
                }
                if (on && l < nl) {
                    double v = cv * (s0 + s1);
                    // round8: cv::max(0) / cv::min(255) / convertTo(CV_8U) (src/filter.cpp:434-436) on the fp64 value --
                    // the plane then holds the 8-bit levels exactly, no fp32 rounding in front of the round-half-even
                    if (round8) v = rint(fmin(255.0, fmax(0.0, v)));
                    sOut[l * W + c] = (float)v;
                }
            }
        };
#pragma unroll
        for (int b = 0; b < kMaxBlocks; ++b) {
            if (4 * b >= wlen) break;
            pixel(idx[b].x & 0xffffu, 4 * b < len);
            NLE_PIXEL_FENCE();
            if (4 * b + 1 < wlen) {
                pixel(idx[b].x >> 16, 4 * b + 1 < len);
                NLE_PIXEL_FENCE();
            }
            if (4 * b + 2 < wlen) {
                pixel(idx[b].y & 0xffffu, 4 * b + 2 < len);
                NLE_PIXEL_FENCE();
            }
            if (4 * b + 3 < wlen) {
                pixel(idx[b].y >> 16, 4 * b + 3 < len);
                NLE_PIXEL_FENCE();
            }
        }
        __syncthreads();  // the row's outputs are in sOut
        for (int l = 0; l < nl; ++l) {
            float* orow = out + (size_t)l * ostride + (size_t)lrow * W;
            for (int c = tid; c < W; c += kT) orow[c] = sOut[l * W + c];
        }
        __syncthreads();  // before the next row writes sOut
    }
}

int sorted_expand_max_cols() { return 36; }
int sorted_expand_max_width() { return 8192; }
// layers one launch takes: four up to 12 columns and 4096 pixels per row, two beyond (registers, LDS)
int sorted_expand_layers(GridSpec gs) { return (gs.nSelCols <= 12 && gs.W <= 4096) ? kExpLayers : 2; }

hipError_t sorted_expand(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                         const double* d_E, const double* d_g, size_t gstride, int nl, const double* d_cvec, float* d_out,
                         long long ostride, bool rec, double kappa, bool round8) {
    const int nC = gs.nSelCols;
    const int lmax = sorted_expand_layers(gs);
    if (nC < 1 || nC > sorted_expand_max_cols() || gs.W > sorted_expand_max_width() || nl < 1 || nl > lmax)
        return hipErrorInvalidValue;
    if (nrows_local <= 0) return hipSuccess;
    const size_t shm = (size_t)((gs.W + 2) & ~1) * sizeof(double) + (size_t)lmax * gs.W * sizeof(float);
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = std::max(1, std::min(nrows_local, ncu));  // one 512-thread workgroup per CU (registers, LDS)
#define NLE_SX1(NCV, LV, RECV)                                                                                         \
    {                                                                                                                  \
        if (shm > 48 * 1024) {                                                                                         \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_expand<NCV, LV, RECV>),         \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                 \
            if (ea != hipSuccess) return ea;                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((k_sorted_expand<NCV, LV, RECV>), dim3((unsigned)grid), dim3(kT), shm, s, d_scol, d_desc,   \
                           gs, nrows_local, d_E, d_g, gstride, nl, d_cvec, d_out, ostride, kappa, round8 ? 1 : 0);     \
    }
#define NLE_SX(NCV)                                                                                                    \
    case NCV:                                                                                                          \
        if constexpr ((NCV) <= 12) {                                                                                   \
            if (lmax == kExpLayers) NLE_SX1(NCV, kExpLayers, false) else NLE_SX1(NCV, 2, false)                        \
        } else {                                                                                                       \
            if (rec) NLE_SX1(NCV, 2, true) else NLE_SX1(NCV, 2, false)                                                 \
        }                                                                                                              \
        break;
    switch (nC) {
        NLE_SX(1) NLE_SX(2) NLE_SX(3) NLE_SX(4) NLE_SX(5) NLE_SX(6) NLE_SX(7) NLE_SX(8) NLE_SX(9) NLE_SX(10) NLE_SX(11)
        NLE_SX(12) NLE_SX(13) NLE_SX(14) NLE_SX(15) NLE_SX(16) NLE_SX(17) NLE_SX(18) NLE_SX(19) NLE_SX(20) NLE_SX(21)
        NLE_SX(22) NLE_SX(23) NLE_SX(24) NLE_SX(25) NLE_SX(26) NLE_SX(27) NLE_SX(28) NLE_SX(29) NLE_SX(30) NLE_SX(31)
        NLE_SX(32) NLE_SX(33) NLE_SX(34) NLE_SX(35) NLE_SX(36)
        default: return hipErrorInvalidValue;
    }
#undef NLE_SX
#undef NLE_SX1
    return hipGetLastError();
}

// ------------------------------------------------------------------ Gram, per-row pair tables (nC <= 11: one launch)
// A_r[(b, b')][x] = sum_{i in row r, x_i = x} c_i^2 ec[c_i][b] ec[c_i][b'],  b <= b': what k_ghist_rows computes with
// nC (nC + 1) / 2 LDS atomics per pixel.  Output layout [row][pair][level], as k_ghist_gemm / k_ghist_final expect.
template <int NC, bool REC>
__global__ __launch_bounds__(kT) void k_sorted_gram(const unsigned short* __restrict__ scol, const uint2* __restrict__ desc,
                                                    const unsigned short* __restrict__ first, GridSpec gs, int nrows,
                                                    const double* __restrict__ Etab, const double* __restrict__ cvec,
                                                    double* __restrict__ Aout, double kappa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NP = NC * (NC + 1) / 2;
    constexpr int SL = 11, PS = SL;  // the tree combines 11 sums at a time (odd stride)
    const int W = gs.W;
    double* sE = reinterpret_cast<double*>(smem_raw);
    double* sP = sE + ((W + 2) & ~1);
    unsigned short* sfirst = reinterpret_cast<unsigned short*>(sP + (size_t)kT * PS);
    const int tid = threadIdx.x;
    for (int i = tid; i <= W; i += kT) sE[i] = Etab[i];
    const int cb0 = gs.colOff, cs = gs.colStep;
    const unsigned sEa = lds_addr(sE);
    const size_t pitch = sorted_row_pitch(W);
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        if (tid < 258) sfirst[tid] = first[(size_t)lrow * 258 + tid];
        const uint2 dsc = desc[(size_t)lrow * kT + tid];
        const int len = dsc_len(dsc);
        double acc[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = 0.0;
        __syncthreads();
        const uint2* slot = reinterpret_cast<const uint2*>(scol + (size_t)lrow * pitch + (size_t)tid * dsc_chp(dsc));
        const double* cv_row = cvec + (size_t)lrow * W;
        uint2 cur = slot[0];
        for (int t0 = 0; t0 < len; t0 += 4) {  // four pixels per step, indices fetched one step ahead (zero padded slots)
            const uint2 nxt = slot[(t0 >> 2) + 1];
            const unsigned c8v[4] = {cur.x & 0xffffu, cur.x >> 16, cur.y & 0xffffu, cur.y >> 16};
            double cfv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cfv[k] = cv_row[c8v[k] >> 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double cf = (t0 + k < len) ? cfv[k] : 0.0;  // padding adds exact zeros
                double q[NC];
                column_factors<NC, REC>(sEa, c8v[k], cb0, cs, kappa, [&](const int b, const double ev) { q[b] = cf * ev; });
                int idx = 0;
#pragma unroll
                for (int b = 0; b < NC; ++b)
#pragma unroll
                    for (int b2 = b; b2 < NC; ++b2) acc[idx++] += q[b] * q[b2];
            }
            cur = nxt;
        }
        const int steps = dsc_steps(dsc), j = dsc_j(dsc), m = dsc_m(dsc);
        double* out = Aout + (size_t)lrow * kLevels * NP;
#pragma unroll
        for (int s0 = 0; s0 < NP; s0 += SL) {
            double v[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i) v[i] = (s0 + i < NP) ? acc[s0 + i] : 0.0;
            combine_chunks<SL, PS>(v, sP, tid, len > 0, j, m, steps);
            const int ns = (NP - s0 < SL) ? NP - s0 : SL;
            for (int i = tid; i < ns * kLevels; i += kT) {
                const int jj = i / kLevels, xx = i & (kLevels - 1);
                const int f0 = sfirst[xx];
                out[(size_t)(s0 + jj) * kLevels + xx] = sfirst[xx + 1] > f0 ? sP[f0 * PS + jj] : 0.0;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ Gram by INDEX SUMS (any nC <= 36: one launch)
// On the reference's sample grid (samplePixels :56-80: equispaced columns g_b = cb0 + b cs) the product of two column factors
// of a pixel depends on the pair (b, b') only through b + b':
//     ec[c][b] ec[c][b'] = exp(-((c - g_b)^2 + (c - g_b')^2) / hx^2) = exp(-(b - b')^2 cs^2 / (2 hx^2)) . G_{b+b'}(c),
//     G_t(c) = exp(-2 (c - m_t)^2 / hx^2),   m_t = cb0 + t cs / 2,   t = 0 .. 2 nC - 2
// (complete the square).  So the nC (nC + 1) / 2 pair tables of k_sorted_gram(_wide) collapse to 2 nC - 1 tables
//     S_r[t][x] = sum_{i in row r, x_i = x} c_i^2 G_t(col_i)
// -- 59 instead of 465 at cfg5, in one launch instead of 15, an eighth of the table bytes -- and the same identity on the rows
// shrinks the GEMM behind it from nR (nR + 1) / 2 to 2 nR - 1 rows (fused.hip: gram_hist).  Exact algebra; only the order and
// grouping of roundings differ from the pair form.
// G_t along t by the recurrence of a Gaussian on an equispaced grid (as column_factors<REC>), started from ONE table value:
//     G_0 = E2[|c - cb0|],   G_{t+1} = G_t rho_t,   rho_0 = exp(theta (c - cb0)) kappa1,   rho_{t+1} = rho_t kappa1^2,
//     E2[d] = exp(-2 d^2 / hx^2),  theta = 2 cs / hx^2,  kappa1 = exp(-cs^2 / (2 hx^2));  exp(theta u) = PH[c >> 6] PL[c & 63].
// The host enables it only where no term leaves fp64's normal range (sorted_gsum_ok); O(t^2) ulp like the other recurrence.
template <int NT>
__global__ __launch_bounds__(kT) void k_sorted_gsum(const unsigned short* __restrict__ scol, const uint2* __restrict__ desc,
                                                    const unsigned short* __restrict__ first, GridSpec gs, int nrows,
                                                    const double* __restrict__ E2tab, const double* __restrict__ cvec,
                                                    double* __restrict__ Aout, double theta, double kappa1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int SL = 11, PS = SL;
    const int W = gs.W;
    double* sE = reinterpret_cast<double*>(smem_raw);               // E2[0 .. W]
    double* sPH = sE + ((W + 2) & ~1);                              // exp(theta (64 k - cb0)), k <= W / 64
    double* sPL = sPH + (((W >> 6) + 2) & ~1);                      // exp(theta l), l < 64
    double* sP = sPL + 64;
    unsigned short* sfirst = reinterpret_cast<unsigned short*>(sP + (size_t)kT * PS);
    const int tid = threadIdx.x;
    const int cb0 = gs.colOff;
    for (int i = tid; i <= W; i += kT) sE[i] = E2tab[i];
    for (int i = tid; i <= (W >> 6); i += kT) sPH[i] = exp(theta * (double)(64 * i - cb0));
    if (tid < 64) sPL[tid] = exp(theta * (double)tid);
    const double kappa2 = kappa1 * kappa1;
    const size_t pitch = sorted_row_pitch(W);
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        if (tid < 258) sfirst[tid] = first[(size_t)lrow * 258 + tid];
        const uint2 dsc = desc[(size_t)lrow * kT + tid];
        const int len = dsc_len(dsc);
        double acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = 0.0;
        __syncthreads();
        const uint2* slot = reinterpret_cast<const uint2*>(scol + (size_t)lrow * pitch + (size_t)tid * dsc_chp(dsc));
        const double* cv_row = cvec + (size_t)lrow * W;
        uint2 cur = slot[0];
        for (int t0 = 0; t0 < len; t0 += 4) {
            const uint2 nxt = slot[(t0 >> 2) + 1];
            const unsigned c8v[4] = {cur.x & 0xffffu, cur.x >> 16, cur.y & 0xffffu, cur.y >> 16};
            double cfv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cfv[k] = cv_row[c8v[k] >> 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double cf = (t0 + k < len) ? cfv[k] : 0.0;  // padding adds exact zeros
                const int c = (int)(c8v[k] >> 3);
                const int du = c - cb0;
                double g = cf * cf * sE[du < 0 ? -du : du];
                double rho = sPH[c >> 6] * sPL[c & 63] * kappa1;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] += g;
                    g *= rho;
                    rho *= kappa2;
                }
            }
            cur = nxt;
        }
        const int steps = dsc_steps(dsc), j = dsc_j(dsc), m = dsc_m(dsc);
        double* out = Aout + (size_t)lrow * kLevels * NT;
#pragma unroll
        for (int s0 = 0; s0 < NT; s0 += SL) {
            double v[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i) v[i] = (s0 + i < NT) ? acc[s0 + i] : 0.0;
            combine_chunks<SL, PS>(v, sP, tid, len > 0, j, m, steps);
            const int ns = (NT - s0 < SL) ? NT - s0 : SL;
            for (int i = tid; i < ns * kLevels; i += kT) {
                const int jj = i / kLevels, xx = i & (kLevels - 1);
                const int f0 = sfirst[xx];
                out[(size_t)(s0 + jj) * kLevels + xx] = sfirst[xx + 1] > f0 ? sP[f0 * PS + jj] : 0.0;
            }
            __syncthreads();
        }
    }
}

// where the recurrence above stays inside fp64's normal range (with a wide margin): every G_t and every rho_t, and the
// exp(theta u) table; the same on the rows (the GEMM's F[r][s] is evaluated directly, no recurrence)
bool sorted_gsum_ok(GridSpec gs, double hx) {
    if (std::getenv("NLE_GRAM_PAIRS") != nullptr) return false;  // measurement: the pair-table form
    const double cs = gs.colStep, nC = gs.nSelCols, Wd = gs.W;
    // every centre m_t lies inside the image, so |c - m_t| < W: -log of the smallest G_t; and the |log| of the largest
    // rho_t = exp(theta (c - cb0) - (2 t + 1) cs^2 / (2 hx^2)) and of the exp(theta (64 k - cb0)) table
    const double m_g = 2.0 * Wd * Wd / (hx * hx);
    const double m_rho = (2.0 * cs * Wd + 2.0 * nC * cs * cs) / (hx * hx);
    return gs.W <= sorted_max_width() && m_g < 600.0 && m_rho < 600.0;
}

hipError_t sorted_gram_sums(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                            const unsigned short* d_first, const double* d_E2, const double* d_cvec, double* d_Aout,
                            double hx) {
    const int nC = gs.nSelCols, nt = 2 * nC - 1;
    if (nC < 1 || nC > 36 || gs.W > sorted_max_width()) return hipErrorInvalidValue;
    if (nrows_local <= 0) return hipSuccess;
    const double cs = gs.colStep;
    const double theta = 2.0 * cs / (hx * hx), kappa1 = std::exp(-cs * cs / (2.0 * hx * hx));
    const size_t shm = sorted_lds_bytes(gs.W, 11) + (size_t)((((gs.W >> 6) + 2) & ~1) + 64) * sizeof(double);
    const int grid = sorted_grid(nrows_local);
#define NLE_GS(NTV)                                                                                                      \
    case NTV: {                                                                                                          \
        if (shm > 48 * 1024) {                                                                                           \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_gsum<NTV>),                       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                   \
            if (ea != hipSuccess) return ea;                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((k_sorted_gsum<NTV>), dim3((unsigned)grid), dim3(kT), shm, s, d_scol, d_desc, d_first, gs,     \
                           nrows_local, d_E2, d_cvec, d_Aout, theta, kappa1);                                            \
    } break;
    switch (nt) {
        NLE_GS(1) NLE_GS(3) NLE_GS(5) NLE_GS(7) NLE_GS(9) NLE_GS(11) NLE_GS(13) NLE_GS(15) NLE_GS(17) NLE_GS(19) NLE_GS(21)
        NLE_GS(23) NLE_GS(25) NLE_GS(27) NLE_GS(29) NLE_GS(31) NLE_GS(33) NLE_GS(35) NLE_GS(37) NLE_GS(39) NLE_GS(41)
        NLE_GS(43) NLE_GS(45) NLE_GS(47) NLE_GS(49) NLE_GS(51) NLE_GS(53) NLE_GS(55) NLE_GS(57) NLE_GS(59) NLE_GS(61)
        NLE_GS(63) NLE_GS(65) NLE_GS(67) NLE_GS(69) NLE_GS(71)
        default: return hipErrorInvalidValue;
    }
#undef NLE_GS
    return hipGetLastError();
}

// ------------------------------------------------------------------ Gram pair tables for wider grids (12 <= nC <= 36)
// nC (nC + 1) / 2 accumulators do not fit in registers any more, so a launch takes BB = 64 / nC rows b0 .. b0 + nb - 1 of
// the pair triangle: acc[i][b'] += q_{b0+i} q_{b'} for ALL b' (the b' < b half is computed and dropped: the row index b0 is
// a run-time value and register arrays need static indices), the column factors by recurrence where the host allows it.
// ceil(nC / BB) launches, each a pass over the sorted pixels; still no atomics: the pair tables of a 10 x 20 grid (most of
// the reference's README runs) are bitwise reproducible like the rest.
__host__ __device__ constexpr int gram_wide_bb(int nc) { return 64 / nc > 0 ? 64 / nc : 1; }

template <int NC>
__global__ __launch_bounds__(kT) void k_sorted_gram_wide(const unsigned short* __restrict__ scol, const uint2* __restrict__ desc,
                                                         const unsigned short* __restrict__ first, GridSpec gs, int nrows,
                                                         const double* __restrict__ Etab, const double* __restrict__ cvec,
                                                         double* __restrict__ Aout, int rec, double kappa, int b0, int nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NP = NC * (NC + 1) / 2;
    constexpr int BB = gram_wide_bb(NC), NV = BB * NC;
    constexpr int SL = 11, PS = SL;
    const int W = gs.W;
    double* sE = reinterpret_cast<double*>(smem_raw);
    double* sP = sE + ((W + 2) & ~1);
    unsigned short* sfirst = reinterpret_cast<unsigned short*>(sP + (size_t)kT * PS);
    const int tid = threadIdx.x;
    for (int i = tid; i <= W; i += kT) sE[i] = Etab[i];
    const int cb0 = gs.colOff, cs = gs.colStep;
    const unsigned sEa = lds_addr(sE);
    const size_t pitch = sorted_row_pitch(W);
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        if (tid < 258) sfirst[tid] = first[(size_t)lrow * 258 + tid];
        const uint2 dsc = desc[(size_t)lrow * kT + tid];
        const int len = dsc_len(dsc);
        double acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.0;
        __syncthreads();
        const uint2* slot = reinterpret_cast<const uint2*>(scol + (size_t)lrow * pitch + (size_t)tid * dsc_chp(dsc));
        const double* cv_row = cvec + (size_t)lrow * W;
        uint2 cur = slot[0];
        for (int t0 = 0; t0 < len; t0 += 4) {  // four pixels per step, indices fetched one step ahead (zero padded slots)
            const uint2 nxt = slot[(t0 >> 2) + 1];
            double cfv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cfv[k] = cv_row[((((k & 2) ? cur.y : cur.x) >> ((k & 1) << 4)) & 0xffffu) >> 3];
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const unsigned c8 = (((k & 2) ? cur.y : cur.x) >> ((k & 1) << 4)) & 0xffffu;
                double cf = (k == 0) ? cfv[0] : (k == 1) ? cfv[1] : (k == 2) ? cfv[2] : cfv[3];
                if (t0 + k >= len) cf = 0.0;  // padding adds exact zeros
                double q[NC], qs[BB];
#pragma unroll
                for (int i = 0; i < BB; ++i) qs[i] = 0.0;
                auto take = [&](const int b, const double ev) {
                    q[b] = cf * ev;
#pragma unroll
                    for (int i = 0; i < BB; ++i)
                        if (b == b0 + i) qs[i] = q[b];  // uniform
                };
                if (rec) column_factors<NC, true>(sEa, c8, cb0, cs, kappa, take);
                else column_factors<NC, false>(sEa, c8, cb0, cs, kappa, take);
#pragma unroll
                for (int i = 0; i < BB; ++i)
#pragma unroll
                    for (int b2 = 0; b2 < NC; ++b2) acc[i * NC + b2] += qs[i] * q[b2];
            }
            cur = nxt;
        }
        const int steps = dsc_steps(dsc), j = dsc_j(dsc), m = dsc_m(dsc);
        double* out = Aout + (size_t)lrow * kLevels * NP;
#pragma unroll
        for (int s0 = 0; s0 < NV; s0 += SL) {
            double v[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i) v[i] = (s0 + i < NV) ? acc[s0 + i] : 0.0;
            combine_chunks<SL, PS>(v, sP, tid, len > 0, j, m, steps);
            const int ns = (NV - s0 < SL) ? NV - s0 : SL;
            for (int i = tid; i < ns * kLevels; i += kT) {
                const int jj = i / kLevels, xx = i & (kLevels - 1);
                const int f = s0 + jj, bi = f / NC, b2 = f - bi * NC, b = b0 + bi;
                if (bi < nb && b2 >= b) {  // pair (b, b2), b <= b2, at its place in the triangle
                    const int pair = b * NC - (b * (b - 1)) / 2 + (b2 - b);
                    const int f0 = sfirst[xx];
                    out[(size_t)pair * kLevels + xx] = sfirst[xx + 1] > f0 ? sP[f0 * PS + jj] : 0.0;
                }
            }
            __syncthreads();
        }
    }
}

int sorted_gram_max_cols() { return 36; }


hipError_t sorted_gram_rows(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                            const unsigned short* d_first, const double* d_E, const double* d_cvec, double* d_Aout, bool rec,
                            double kappa) {
    const int nC = gs.nSelCols;
    if (nC < 1 || nC > sorted_gram_max_cols() || gs.W > sorted_max_width()) return hipErrorInvalidValue;
    if (nrows_local <= 0) return hipSuccess;
    const size_t shm = sorted_lds_bytes(gs.W, 11);
    const int grid = sorted_grid(nrows_local);
    if (nC > 11) {
        const int bb = gram_wide_bb(nC);
#define NLE_SGW(NCV)                                                                                                    \
    case NCV: {                                                                                                         \
        if (shm > 48 * 1024) {                                                                                          \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_gram_wide<NCV>),                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                  \
            if (ea != hipSuccess) return ea;                                                                            \
        }                                                                                                               \
        for (int b0 = 0; b0 < nC; b0 += bb)                                                                             \
            hipLaunchKernelGGL((k_sorted_gram_wide<NCV>), dim3((unsigned)grid), dim3(kT), shm, s, d_scol, d_desc,       \
                               d_first, gs, nrows_local, d_E, d_cvec, d_Aout, rec ? 1 : 0, kappa, b0,                    \
                               std::min(bb, nC - b0));                                                                  \
    } break;
        switch (nC) {
            NLE_SGW(12) NLE_SGW(13) NLE_SGW(14) NLE_SGW(15) NLE_SGW(16) NLE_SGW(17) NLE_SGW(18) NLE_SGW(19) NLE_SGW(20)
            NLE_SGW(21) NLE_SGW(22) NLE_SGW(23) NLE_SGW(24) NLE_SGW(25) NLE_SGW(26) NLE_SGW(27) NLE_SGW(28) NLE_SGW(29)
            NLE_SGW(30) NLE_SGW(31) NLE_SGW(32) NLE_SGW(33) NLE_SGW(34) NLE_SGW(35) NLE_SGW(36)
            default: return hipErrorInvalidValue;
        }
#undef NLE_SGW
        return hipGetLastError();
    }
#define NLE_SG1(NCV, RECV)                                                                                              \
    {                                                                                                                   \
        if (shm > 48 * 1024) {                                                                                          \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_gram<NCV, RECV>),                \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                  \
            if (ea != hipSuccess) return ea;                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((k_sorted_gram<NCV, RECV>), dim3((unsigned)grid), dim3(kT), shm, s, d_scol, d_desc, d_first,  \
                           gs, nrows_local, d_E, d_cvec, d_Aout, kappa);                                                \
    }
#define NLE_SG(NCV)                                                                                                     \
    case NCV:                                                                                                           \
        if (rec) return hipErrorInvalidValue; /* up to 11 columns: table form only */                                    \
        NLE_SG1(NCV, false)                                                                                             \
        break;
    switch (nC) {
        NLE_SG(1) NLE_SG(2) NLE_SG(3) NLE_SG(4) NLE_SG(5) NLE_SG(6) NLE_SG(7) NLE_SG(8) NLE_SG(9) NLE_SG(10) NLE_SG(11)
        default: return hipErrorInvalidValue;
    }
#undef NLE_SG
#undef NLE_SG1
    return hipGetLastError();
}

}  // namespace nlek
