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
// Reference arithmetic restated: the Sinkhorn row products / column sums of src/filter.cpp:238-245, the Gram
// Wab Wab^T of :296 and the reduce half of apply (:456), exactly as fused.hip derives them; only the order of the
// fp64 sums differs.
#include "kernels.h"

#include <algorithm>

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

// byte offset of E[|c - c_b|] in the LDS table from the pre-scaled 16-bit operands c8 = 8 c, cb8 = 8 c_b (8 W <= 65536):
// one v_sad_u16 instead of subtract / negate / max / shift
__device__ __forceinline__ double e_at(const double* sE, unsigned c8, unsigned cb8) {
    const unsigned off = __builtin_amdgcn_sad_u16(c8, cb8, 0u);
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sE) + off);
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

int sorted_max_width() { return 8192; }

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
// One workgroup (256 threads) per local image row.  Sample pixels are left out (the N-sized sums skip them: their Phi
// rows are the exact V_A rows, reference :275).  Outputs, per row:
//   scol[W]    8 x column (the byte offset the pass kernels feed to v_sad_u16) in (level, column) order -- a STABLE
//              counting sort, so that the summation order of every later pass is a function of the image alone;
//   desc[kT]   one chunk per pass thread: x = start | stride << 16, y = len | level << 16 (len 0: idle thread);
//   first[258] first[x] = first chunk of level x (x = 0..256), first[257] = number of tree steps = ceil(log2(max m)).
__global__ __launch_bounds__(kSortThreads) void k_sort_rows(const float* __restrict__ lum, GridSpec gs, int row0,
                                                            unsigned short* __restrict__ scol, uint2* __restrict__ desc,
                                                            unsigned short* __restrict__ first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W = gs.W;
    unsigned short* srow = reinterpret_cast<unsigned short*>(smem_raw);  // [W] sorted columns
    __shared__ int tot[kLevels], off[kLevels + 1], run[kLevels], fch[kLevels + 1], red[8];
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
    const int CH = lo;
    const int m = (mine + CH - 1) / CH;
    const int maxm = block_max256(m, red);
    fch[tid + 1] = m;
    if (tid == 0) fch[0] = 0;
    __syncthreads();
    for (int d = 1; d < kLevels; d <<= 1) {
        const int v = (tid + 1 > d) ? fch[tid + 1 - d] : 0;
        __syncthreads();
        if (tid + 1 > d) fch[tid + 1] += v;
        __syncthreads();
    }
    unsigned short* frow = first + (size_t)lrow * 258;
    frow[tid] = (unsigned short)fch[tid];
    if (tid == 0) {
        frow[kLevels] = (unsigned short)fch[kLevels];
        int steps = 0;
        while ((1 << steps) < maxm) ++steps;
        frow[kLevels + 1] = (unsigned short)steps;
    }
    const int nchunks = fch[kLevels];
    for (int k = tid; k < kT; k += kSortThreads) {
        uint2 d = make_uint2(1u << 16, 0u);  // idle: stride 1, len 0
        if (k < nchunks) {
            int a = 0, b = kLevels;  // largest x with fch[x] <= k
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (fch[mid] <= k) a = mid;
                else b = mid;
            }
            const int x = a, j = k - fch[x], mx = fch[x + 1] - fch[x], cnt = tot[x];
            d = make_uint2((unsigned)(off[x] + j) | ((unsigned)mx << 16), (unsigned)((cnt - j + mx - 1) / mx) | ((unsigned)x << 16));
        }
        desc[(size_t)lrow * kT + k] = d;
    }
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
            srow[off[x] + run[x] + prev + rank] = (unsigned short)(c << 3);  // pre-scaled: 8 c (W <= 8192)
        }
        __syncthreads();
        run[tid] += cntG[0][tid] + cntG[1][tid] + cntG[2][tid] + cntG[3][tid];
        __syncthreads();
    }
    unsigned short* out = scol + (size_t)lrow * W;
    for (int i = tid; i < W; i += kSortThreads) out[i] = i < wn ? srow[i] : (unsigned short)0;
}

// elements of the scol buffer: the pass kernels prefetch column indices up to two chunk strides (<= 2 x 512 entries)
// past a thread's chunk, i.e. past the last row's end
size_t sorted_scol_elems(int W, int nrows_local) { return (size_t)std::max(nrows_local, 0) * W + 2 * kSortedThreads + 8; }

hipError_t sort_rows(hipStream_t s, const float* d_lum, GridSpec gs, int row0, int nrows_local, unsigned short* d_scol,
                     uint2* d_desc, unsigned short* d_first) {
    if (gs.W > sorted_max_width() || nrows_local <= 0) return nrows_local <= 0 ? hipSuccess : hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)nrows_local), dim3(kSortThreads), (size_t)gs.W * sizeof(unsigned short), s,
                       d_lum, gs, row0, d_scol, d_desc, d_first);
    return hipGetLastError();
}

// ------------------------------------------------------------------ LDS layout shared by the two pass kernels
// sE [W + 1] doubles | sP [kT][PS] doubles | sfirst [258] u16
__host__ __device__ inline size_t sorted_lds_bytes(int W, int ps) {
    return ((size_t)((W + 2) & ~1) + (size_t)kT * ps) * sizeof(double) + 260 * sizeof(unsigned short);
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
template <int NC>
__global__ __launch_bounds__(kT) void k_sorted_pass(int mode, const unsigned short* __restrict__ scol,
                                                    const uint2* __restrict__ desc, const unsigned short* __restrict__ first,
                                                    GridSpec gs, int row0, int nrows, const double* __restrict__ Etab,
                                                    const double* __restrict__ g, double eps, double* __restrict__ ybuf,
                                                    double* __restrict__ hout, const double* __restrict__ cvec,
                                                    const float* __restrict__ xvec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int n = kLevels * NC;
    constexpr int SL = NC < 11 ? NC : 11;  // sums combined per tree (slices of the nC sums when nC > 11)
    constexpr int PS = SL | 1;             // odd stride: consecutive threads' rows start on different banks
    constexpr bool KEEP_E = NC <= 12;      // keep the column factors of a pixel in registers between the two loops
    const int W = gs.W;
    double* sE = reinterpret_cast<double*>(smem_raw);
    double* sP = sE + ((W + 2) & ~1);
    unsigned short* sfirst = reinterpret_cast<unsigned short*>(sP + (size_t)kT * PS);
    const int tid = threadIdx.x;
    for (int i = tid; i <= W; i += kT) sE[i] = Etab[i];
    const int cb0 = gs.colOff, cs = gs.colStep;
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        if (tid < 258) sfirst[tid] = first[(size_t)lrow * 258 + tid];
        const uint2 dsc = desc[(size_t)lrow * kT + tid];
        const int start = (int)(dsc.x & 0xffffu), stride = (int)(dsc.x >> 16), len = (int)(dsc.y & 0xffffu),
                  x = (int)(dsc.y >> 16);
        double gv[NC], acc[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            gv[b] = (mode == ROWPASS_RECIP && len > 0) ? g[(size_t)lrow * n + b * kLevels + x] : 0.0;
            acc[b] = 0.0;
        }
        __syncthreads();  // sE / sfirst visible; the previous row's reads of sP are done
        // software pipeline: the (dependent) loads of pixel t + 1 -- its column index, and in the apply its c and x --
        // are issued, unconditionally, before pixel t is processed
        const unsigned short* sc = scol + (size_t)lrow * W + start;
        const double* cv_row = cvec ? cvec + (size_t)lrow * W : nullptr;
        const float* xv_row = xvec ? xvec + (size_t)(row0 + lrow) * W : nullptr;
        // (the index loads run up to two chunk strides past the chunk: scol is padded for that, sort_rows_elems)
        unsigned c8 = sc[0], c8n = sc[stride];
        double yx = (mode == ROWPASS_XVEC) ? cv_row[c8 >> 3] * (double)xv_row[c8 >> 3] : 1.0;
        for (int t = 0; t < len; ++t) {
            const unsigned c8nn = sc[(t + 2) * stride];
            const double yxn = (mode == ROWPASS_XVEC) ? cv_row[c8n >> 3] * (double)xv_row[c8n >> 3] : 1.0;
            double e[KEEP_E ? NC : 1];
            double y = 1.0;
            if (mode == ROWPASS_RECIP) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int b = 0; b < NC; ++b) {
                    const double ev = e_at(sE, c8, (unsigned)(cb0 + b * cs) << 3);
                    if constexpr (KEEP_E) e[b] = ev;
                    if (b & 1) s1 += ev * gv[b];
                    else s0 += ev * gv[b];
                }
                y = recip0_d(s0 + s1, eps);
            } else {
                if constexpr (KEEP_E) {
#pragma unroll
                    for (int b = 0; b < NC; ++b) e[b] = e_at(sE, c8, (unsigned)(cb0 + b * cs) << 3);
                }
                if (mode == ROWPASS_XVEC) y = yx;  // apply: y_i = c_i x_i
            }
            if (ybuf != nullptr) ybuf[(size_t)lrow * W + (c8 >> 3)] = y;
#pragma unroll
            for (int b = 0; b < NC; ++b) {
                if constexpr (KEEP_E) acc[b] += e[b] * y;
                else acc[b] += e_at(sE, c8, (unsigned)(cb0 + b * cs) << 3) * y;
            }
            c8 = c8n;
            c8n = c8nn;
            yx = yxn;
        }
        const int steps = sfirst[kLevels + 1], j = tid - sfirst[x], m = stride;
        double* hrow = hout + (size_t)lrow * n;
#pragma unroll
        for (int s0 = 0; s0 < NC; s0 += SL) {
            double v[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i) v[i] = (s0 + i < NC) ? acc[s0 + i] : 0.0;
            combine_chunks<SL, PS>(v, sP, tid, len > 0, j, m, steps);
            const int ns = (NC - s0 < SL) ? NC - s0 : SL;
            for (int i = tid; i < ns * kLevels; i += kT) {
                const int bb = i / kLevels, xx = i & (kLevels - 1);
                const int f0 = sfirst[xx];
                hrow[(size_t)(s0 + bb) * kLevels + xx] = sfirst[xx + 1] > f0 ? sP[f0 * PS + bb] : 0.0;
            }
            __syncthreads();  // before the next slice / row overwrites sP and sfirst
        }
    }
}

static int sorted_grid(int nrows) {
    int ncu = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    return std::max(1, std::min(nrows, 2 * ncu));  // two 512-thread workgroups per CU, each walks its rows
}

hipError_t sorted_pass(hipStream_t s, int mode, GridSpec gs, int row0, int nrows_local, const unsigned short* d_scol,
                       const uint2* d_desc, const unsigned short* d_first, const double* d_E, const double* d_g, double eps,
                       double* d_ybuf, double* d_h, const double* d_cvec, const float* d_xvec) {
    const int nC = gs.nSelCols;
    if (nC < 1 || nC > 36 || gs.W > sorted_max_width()) return hipErrorInvalidValue;
    if (nrows_local <= 0) return hipSuccess;
    const size_t shm = sorted_lds_bytes(gs.W, (nC < 11 ? nC : 11) | 1);
    const int grid = sorted_grid(nrows_local);
#define NLE_SP(NCV)                                                                                                       \
    case NCV: {                                                                                                           \
        if (shm > 48 * 1024) {                                                                                            \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_pass<NCV>),                        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                    \
            if (ea != hipSuccess) return ea;                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_sorted_pass<NCV>), dim3((unsigned)grid), dim3(kT), shm, s, mode, d_scol, d_desc, d_first, gs, \
                           row0, nrows_local, d_E, d_g, eps, d_ybuf, d_h, d_cvec, d_xvec);                                \
    } break;
    switch (nC) {
        NLE_SP(1) NLE_SP(2) NLE_SP(3) NLE_SP(4) NLE_SP(5) NLE_SP(6) NLE_SP(7) NLE_SP(8) NLE_SP(9) NLE_SP(10) NLE_SP(11)
        NLE_SP(12) NLE_SP(13) NLE_SP(14) NLE_SP(15) NLE_SP(16) NLE_SP(17) NLE_SP(18) NLE_SP(19) NLE_SP(20)
        NLE_SP(21) NLE_SP(22) NLE_SP(23) NLE_SP(24) NLE_SP(25) NLE_SP(26) NLE_SP(27) NLE_SP(28) NLE_SP(29)
        NLE_SP(30) NLE_SP(31) NLE_SP(32) NLE_SP(33) NLE_SP(34) NLE_SP(35) NLE_SP(36)
        default: return hipErrorInvalidValue;
    }
#undef NLE_SP
    return hipGetLastError();
}

// ------------------------------------------------------------------ Gram, per-row pair tables (nC <= 11)
// A_r[(b, b')][x] = sum_{i in row r, x_i = x} c_i^2 ec[c_i][b] ec[c_i][b'],  b <= b': what k_ghist_rows computes with
// nC (nC + 1) / 2 LDS atomics per pixel.  Output layout [row][pair][level], as k_ghist_gemm / k_ghist_final expect.
template <int NC>
__global__ __launch_bounds__(kT) void k_sorted_gram(const unsigned short* __restrict__ scol, const uint2* __restrict__ desc,
                                                    const unsigned short* __restrict__ first, GridSpec gs, int nrows,
                                                    const double* __restrict__ Etab, const double* __restrict__ cvec,
                                                    double* __restrict__ Aout) {
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
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        if (tid < 258) sfirst[tid] = first[(size_t)lrow * 258 + tid];
        const uint2 dsc = desc[(size_t)lrow * kT + tid];
        const int start = (int)(dsc.x & 0xffffu), stride = (int)(dsc.x >> 16), len = (int)(dsc.y & 0xffffu),
                  x = (int)(dsc.y >> 16);
        double acc[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[i] = 0.0;
        __syncthreads();
        const unsigned short* sc = scol + (size_t)lrow * W + start;
        const double* cv_row = cvec + (size_t)lrow * W;
        unsigned c8 = sc[0], c8n = sc[stride];
        double cf = cv_row[c8 >> 3];
        for (int t = 0; t < len; ++t) {  // loads of pixel t + 1 (and the index of t + 2) in flight under pixel t
            const unsigned c8nn = sc[(t + 2) * stride];
            const double cfn = cv_row[c8n >> 3];
            double q[NC];
#pragma unroll
            for (int b = 0; b < NC; ++b) q[b] = cf * e_at(sE, c8, (unsigned)(cb0 + b * cs) << 3);
            int idx = 0;
#pragma unroll
            for (int b = 0; b < NC; ++b)
#pragma unroll
                for (int b2 = b; b2 < NC; ++b2) acc[idx++] += q[b] * q[b2];
            c8 = c8n;
            c8n = c8nn;
            cf = cfn;
        }
        const int steps = sfirst[kLevels + 1], j = tid - sfirst[x], m = stride;
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

int sorted_gram_max_cols() { return 11; }

hipError_t sorted_gram_rows(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                            const unsigned short* d_first, const double* d_E, const double* d_cvec, double* d_Aout) {
    const int nC = gs.nSelCols;
    if (nC < 1 || nC > sorted_gram_max_cols() || gs.W > sorted_max_width()) return hipErrorInvalidValue;
    if (nrows_local <= 0) return hipSuccess;
    const size_t shm = sorted_lds_bytes(gs.W, 11);
    const int grid = sorted_grid(nrows_local);
#define NLE_SG(NCV)                                                                                                     \
    case NCV: {                                                                                                         \
        if (shm > 48 * 1024) {                                                                                          \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sorted_gram<NCV>),                      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                  \
            if (ea != hipSuccess) return ea;                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((k_sorted_gram<NCV>), dim3((unsigned)grid), dim3(kT), shm, s, d_scol, d_desc, d_first, gs,   \
                           nrows_local, d_E, d_cvec, d_Aout);                                                           \
    } break;
    switch (nC) {
        NLE_SG(1) NLE_SG(2) NLE_SG(3) NLE_SG(4) NLE_SG(5) NLE_SG(6) NLE_SG(7) NLE_SG(8) NLE_SG(9) NLE_SG(10) NLE_SG(11)
        default: return hipErrorInvalidValue;
    }
#undef NLE_SG
    return hipGetLastError();
}

}  // namespace nlek
