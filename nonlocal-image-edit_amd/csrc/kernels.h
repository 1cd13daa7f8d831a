// Launch wrappers of the gfx950 kernels (kernels.hip).  Internal to libnle_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace nlek {

// Closed form of samplePixels (reference src/filter.cpp:56-80).
struct GridSpec {
    int H, W;
    int rowStep, rowOff, nSelRows;
    int colStep, colOff, nSelCols;
    __host__ __device__ int p() const { return nSelRows * nSelCols; }
};

// {row, col, luminance, 0} of one sample, fp32 (coordinates are exact in fp32)
typedef float4 Sample4;  // x = row, y = col, z = luminance, w = 0

constexpr int kRowpassMaxBlocks = 1024;
constexpr int kGramTilesPerWave = 7;
constexpr int kGramRowsPerStage = 32;

enum RowpassMode { ROWPASS_COLSUM = 0, ROWPASS_RECIP = 1, ROWPASS_XVEC = 2 };

// out[k] = lum[sel_index(k)] for the p samples
hipError_t gather_samples(hipStream_t s, const float* d_lum, GridSpec gs, float* d_out);

// K_AB rows, natural order: kab[i][s] = exp(-(sw*d2 + pw*dv^2)), i in [pix0, pix0+M)
hipError_t affinity(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples,
                    int p, int ld, float sw, float pw, long long pix0, long long M,
                    float* d_kab);

// C (M x ldc) = rowscale o (A x B).  B: kd x ldb fp32 row-major (zero padded to ldc cols).
// fused != 0: A rows are affinities computed on the fly from (lum, samples); else A is
// read from d_A (M x lda).  d_u != null (non-fused): rowscale_i = recip(A_i . u).
hipError_t ts_gemm(hipStream_t s, bool fused, const float* d_A, int lda, const float* d_lum,
                   GridSpec gs, const Sample4* d_samples, float sw, float pw, long long pix0,
                   const float* d_B, int ldb, int kd, float* d_C, int ldc, long long M,
                   const double* d_u, double eps);

// One pass over X (M x ld): partial[b][j] = sum_{rows of block b} X[i][j] * y_i,
//   mode COLSUM: y=1; RECIP: y_i = recip(X_i . (lam o t_in)); XVEC: y_i = xvec[i].
// Returns the number of blocks used in *nblocks.
hipError_t rowpass(hipStream_t s, int mode, const float* d_X, long long M, int ld,
                   const double* d_t_in, const double* d_lam, const float* d_xvec, double eps,
                   double* d_partial, int* nblocks);
// t_out[j] = sum_b partial[b][j], j < ld
hipError_t reduce_partials(hipStream_t s, const double* d_partial, int nblocks, int ld,
                           double* d_t_out);
// u[j] = lam[j] * t[j]
hipError_t scale_vec(hipStream_t s, const double* d_lam, const double* d_t, int n, double* d_u);
// out[i] = recip(X_i . u)
hipError_t row_scalings(hipStream_t s, const float* d_X, long long M, int ld, const double* d_u,
                        double eps, double* d_out);

// Gram: tiles of G = sum_i c_i^2 x_i x_i^T, c_i = recip(x_i . u); upper-triangular 32x32
// tiles, chunked fp32 MFMA accumulation, fp64 across chunks.
// workspace: d_partial [nchunks][ntiles][1024] doubles; result d_tiles [ntiles][1024].
int gram_num_tiles(int ld);
int gram_chunk_rows(long long M);
size_t gram_partial_elems(long long M, int ld);
hipError_t gram(hipStream_t s, const float* d_X, long long M, int ld, const double* d_u,
                double eps, double* d_partial, double* d_tiles);

// Y[l][i] = sum_k V[i][k] * g[l][k]   (g: L x ld doubles, device)
hipError_t apply_expand(hipStream_t s, const float* d_V, long long M, int ld, const double* d_g,
                        int L, float* d_Y, long long ystride);

// X[idx[k]] = src[k] (ld floats each) for idx[k] in [0, M)
hipError_t scatter_rows(hipStream_t s, const float* d_src, const long long* d_idx, int n, int ld,
                        float* d_X, long long M);

}  // namespace nlek
