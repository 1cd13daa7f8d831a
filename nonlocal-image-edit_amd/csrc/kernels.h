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

#if defined(__HIPCC__)
// K(i, s) = exp(negativeWeightedDistance) (reference src/filter.cpp:104-112,144-145) in base 2:
// nsw = -log2(e)/hx^2, npw = -log2(e)/hy^2, one v_exp_f32.  The spatial term is exact in fp32
// (integer coordinates, H*W < 2^24 keeps dr^2+dc^2 < 2^25; it is rounded once above that).
__device__ __forceinline__ float affinity_value(float pr, float pc, float px, Sample4 s, float nsw, float npw) {
    const float dr = pr - s.x, dc = pc - s.y, dv = px - s.z;
    return __builtin_amdgcn_exp2f(nsw * (dr * dr + dc * dc) + npw * (dv * dv));
}
// closed form of the selection predicate of samplePixels (src/filter.cpp:68-70)
__device__ __forceinline__ bool is_sample_pixel(const GridSpec& gs, int row, int col) {
    const int dr = row - gs.rowOff, dc = col - gs.colOff;
    if (dr < 0 || dc < 0) return false;
    const int qr = dr / gs.rowStep, qc = dc / gs.colStep;
    return (qr * gs.rowStep == dr) && (qc * gs.colStep == dc) && qr < gs.nSelRows && qc < gs.nSelCols;
}
#endif

// composite launchers report each kernel they enqueue so that the caller can time them separately
struct LaunchObserver {
    virtual void begin(int sub) = 0;  // sub: index of the kernel inside the composite
    virtual void end() = 0;
    virtual ~LaunchObserver() = default;
};
enum { SUB_HIST_G = 0, SUB_HIST_PIX = 1, SUB_HIST_HH = 2, SUB_HIST_Z = 3 };
enum { SUB_GHIST_ROWS = 0, SUB_GHIST_EE = 1, SUB_GHIST_GEMM = 2, SUB_GHIST_FINAL = 3 };

constexpr int kRowpassMaxBlocks = 1024;
constexpr int kGramTilesPerWave = 7;
constexpr int kGramRowsPerStage = 32;

enum RowpassMode { ROWPASS_COLSUM = 0, ROWPASS_RECIP = 1, ROWPASS_XVEC = 2 };

// out[k] = lum[sel_index(k)] for the p samples
hipError_t gather_samples(hipStream_t s, const float* d_lum, GridSpec gs, float* d_out);
// fp64, 0 for samples outside rows [row0, row1) (d_lum: virtual base of the full image, only those rows exist)
hipError_t gather_samples_slab(hipStream_t s, const float* d_lum, GridSpec gs, int row0, int row1, double* d_out);

// K_AB rows, natural order: kab[i][s] = exp2(nsw*d2 + npw*dv^2), i in [pix0, pix0+M)
hipError_t affinity(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples,
                    int p, int ld, float nsw, float npw, long long pix0, long long M,
                    float* d_kab);

// C (M x ldc) = rowscale o (A x B).  B: kd x ldb fp32 row-major (zero padded to ldc cols).
// fused != 0: A rows are affinities computed on the fly from (lum, samples); else A is
// read from d_A (M x lda).  d_u != null (non-fused): rowscale_i = recip(A_i . u);
// d_c != null (fused): rowscale_i = c_i.
hipError_t ts_gemm(hipStream_t s, bool fused, const float* d_A, int lda, const float* d_lum,
                   GridSpec gs, const Sample4* d_samples, float nsw, float npw, long long pix0,
                   const float* d_B, int ldb, int kd, float* d_C, int ldc, long long M,
                   const double* d_u, double eps, const float* d_c = nullptr);

// The fused form on the bf16 matrix cores with split operands (tsgemm_bf16x3.hip: hi + mid + lo, six products, fp32
// accumulate): d_Bs = ts_gemm_bf16x3_split of the kd x ldb fp32 matrix B (ts_gemm_bf16x3_bsplit_elems shorts)
size_t ts_gemm_bf16x3_bsplit_elems(int kd, int ldb);
hipError_t ts_gemm_bf16x3_split(hipStream_t s, const float* d_B, int kd, int ldb, unsigned short* d_Bs);
hipError_t ts_gemm_bf16x3(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, float nsw, float npw,
                          long long pix0, const unsigned short* d_Bs, int ldb, int kd, float* d_C, int ldc, long long M,
                          const float* d_c = nullptr);

// One pass over X (M x ld): partial[b][j] = sum_{rows of block b} X[i][j] * y_i,
//   mode COLSUM: y=1; RECIP: y_i = recip(X_i . (lam o t_in)); XVEC: y_i = xvec[i].
// Returns the number of blocks used in *nblocks.
hipError_t rowpass(hipStream_t s, int mode, const float* d_X, long long M, int ld,
                   const double* d_t_in, const double* d_lam, const float* d_xvec, double eps,
                   double* d_partial, int* nblocks);
// t_out[sl][j] = sum over the sl-th slice of blocks of partial[b][j], j < ld (nslices rows out;
// nslices == 1: the full column sums)
hipError_t reduce_partials(hipStream_t s, const double* d_partial, int nblocks, int ld,
                           double* d_t_out, int nslices = 1);
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
hipError_t gram_reduce(hipStream_t s, const double* d_partial, int nchunks, int ntiles, double* d_tiles);

// ---- Phi-free ("sample space") passes, fused.hip ----
// Sinkhorn half-iteration without Phi: z[s] = sum_i k_i[s] y_i over non-sample local pixels,
// y_i = 1 (COLSUM) or recip(k_i . w); partial: [sink_pass_rows(M)][sink_pass_ld(p)] doubles.
// d_ybuf (optional): y_i per local pixel, fp64 (0 at sample pixels).  p <= sink_pass_max_p().
int sink_pass_ld(int p);
int sink_pass_rows(long long M);
int sink_pass_max_p();
hipError_t sink_pass(hipStream_t s, int mode, const float* d_lum, GridSpec gs, const Sample4* d_samples,
                     int p, const double* d_w, float nsw, float npw, long long pix0, long long M, double eps,
                     double* d_ybuf, double* d_partial);
// p-sized update between passes in factored form: u' = lambda o (X1^T [z; y_A]), [w'; s_A'] = X2 u' (see fused.hip)
// d_z: zrows x zld partial column sums, added in row order; d_X1: 2p x r column-major, d_X2: 2p x r row-major;
// d_v (2p) and d_u (r): scratch
hipError_t sink_update(hipStream_t s, int mode, int p, int r, bool chol, const double* d_X1, const double* d_X2,
                       const double* d_lam, const double* d_z, int zrows, int zld, const double* d_sA_cur, double eps,
                       double* d_v, double* d_u, double* d_sA_next, double* d_w_next);
// Gk = sum over non-sample local pixels of c_i^2 k_i k_i^T on the fp64 MFMA; upper-triangular
// 16x16 tiles (row-major 256 doubles each), p <= 256.
constexpr int kG64TilesPerWave = 23;
int gram64_ld(int p);
int gram64_num_tiles(int p);
size_t gram64_partial_elems(long long M, int p);
hipError_t gram64(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, int p, float nsw,
                  float npw, long long pix0, long long M, const double* d_c, double* d_partial, double* d_tiles);
// V (M x ldv fp32) = diag(c) K D on the fp64 MFMA; D: p x project64_ld(K) fp64 row-major, K <= 128
int project64_ld(int K);
hipError_t project64(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, int p, float nsw,
                     float npw, long long pix0, long long M, const double* d_D, int K, const double* d_c, float* d_V,
                     int ldv);

// ---- quantised-luminance (integer 0..255) Sinkhorn pass: table look-ups instead of exponentials ----
int sink_hist_max_cols();
hipError_t check_levels(hipStream_t s, const float* d_lum, long long n, int* d_flag);  // 2 ints: [0] != 0: not quantised, [1]: 16-level tiles that occur (bit t)
hipError_t hist_tables(hipStream_t s, GridSpec gs, const Sample4* d_samples, int p, double hx, double hy, int row0,
                       int nrows_local, double* d_er, double* d_ecT, double* d_Ep);
// partial: [nrows_local][ldp] doubles (one row per image row); d_ybuf as in sink_pass
hipError_t sink_hist(hipStream_t s, int mode, const float* d_lum, GridSpec gs, int p, int ldp, int row0,
                     int nrows_local, const double* d_er, const double* d_ecT, const double* d_Ep,
                     const double* d_w, double eps, double* d_ybuf, double* d_partial);

// ---- the literal decomposition in fp64 (generic64.hip): auto mode's fallback and the stage-level API
hipError_t affinity64(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, int p, int ld, double sw,
                      double pw, long long pix0, long long M, double* d_kab, bool skip_samples = false);
hipError_t add64(hipStream_t s, double* d_y, const double* d_x, size_t n);  // y += x
hipError_t row_scalings64(hipStream_t s, const double* d_X, long long M, int ld, int r, const double* d_u, double eps,
                          double* d_out);
// C (M x ldc) = diag(rs) A (M x lda, width kd) B (kd x nc column-major on the DEVICE); rs may be null
hipError_t ts_gemm64(hipStream_t s, const double* d_A, long long M, int lda, int kd, const double* d_B, int nc,
                     const double* d_rs, double* d_C, int ldc);
// C(i,j) = dl[i] (sum_k A(i,k) dk[k] B(k,j)) dr[j] + add(i,j); matrices by (pointer, row stride, column stride)
hipError_t gemm64s(hipStream_t s, int m, int n, int kk, const double* A, long long rsA, long long csA, const double* B,
                   long long rsB, long long csB, double* C, long long rsC, long long csC, const double* dl = nullptr,
                   const double* dk = nullptr, const double* dr = nullptr, const double* add = nullptr, long long rsD = 0,
                   long long csD = 0);
hipError_t rowpass64(hipStream_t s, int mode, const double* d_X, long long M, int ld, const double* d_t_in,
                     const double* d_lam, const float* d_xvec, double eps, double* d_partial, int* nblocks);
// G (r x r, full symmetric) = sum_i cs_i^2 x_i x_i^T (cs null: 1); d_partial: gram64d_partial_elems doubles
size_t gram64d_partial_elems(long long M, int r);
hipError_t gram64d(hipStream_t s, const double* d_X, long long M, int ld, int r, const double* d_cs, double* d_partial,
                   double* d_G);
hipError_t apply_expand64(hipStream_t s, const double* d_V, long long M, int ld, int K, const double* d_g, int L, float* d_Y,
                          long long ystride);
hipError_t scatter_rows64(hipStream_t s, const double* d_src, const long long* d_idx, int n, int ld, double* d_X, long long M);
hipError_t to_f32(hipStream_t s, const double* d_X, long long n, float* d_out);
hipError_t scale_rows64(hipStream_t s, double* d_X, int m, int n, const double* d_dl);  // X (m x n col-major) <- diag(dl) X

// ---- one-workgroup Householder tridiagonalisation (tridiag.hip): Q n x n col-major (lower triangle read), n <= tridiag_max_n();
// outputs in the conventions of nleh::tridiag_reduce (V: u_i in column i rows 0..i-1; hs; d, e)
int tridiag_max_n();
hipError_t tridiag(hipStream_t s, int n, const double* d_Q, const double* d_diag_add, double* d_V, double* d_d, double* d_e,
                   double* d_hs);

// ---- dense fp64 algebra of order n <= sytrd_max_n() on the device (dense64.hip)
// Householder tridiagonalisation in one persistent launch of G workgroups (0: sytrd_groups(n)); d_A n x n column-major, lower
// triangle read, d_diag_add (n, optional) added to the diagonal; d_pub: sytrd_pub_elems(n) doubles (the Householder vectors
// and scales stay there for sytrd_back); d_d[i] = T(i,i), d_e[i] = T(i,i-1), d_e[0] = 0; *d_status != 0 afterwards: a
// hand-off between workgroups timed out (the outputs are then meaningless)
int sytrd_max_n();
int sytrd_groups(int n);
size_t sytrd_pub_elems(int n);
hipError_t sytrd_dist(hipStream_t s, int n, int G, const double* d_A, const double* d_diag_add, double* d_pub, double* d_d,
                      double* d_e, int* d_status);
// all n eigenvalues of the tridiagonal matrix, DESCENDING, by multi-section Sturm counts (absolute accuracy ~ulp ||T||)
hipError_t tridiag_bisect(hipStream_t s, int n, const double* d_d, const double* d_e, double* d_D);
// Z (n x K, column stride ldz) <- Q Z with the orthogonal factor of sytrd_dist's reduction
hipError_t sytrd_back(hipStream_t s, int n, const double* d_pub, int K, double* d_Z, int ldz);
// Cholesky factor L of d_A (lower triangle read; d_A untouched), L^-1 and d_scal[0] = trace(A^-1); *d_status |= 2 when a pivot
// is not positive; d_tmp: potrf_tmp_elems(n) doubles of scratch
size_t potrf_tmp_elems(int n);
hipError_t potrf_inverse(hipStream_t s, int n, const double* d_A, double* d_L, double* d_Linv, double* d_tmp, double* d_scal,
                         int* d_status);
// d_dst = d_src^T, n x n column-major with column strides lds / ldd (0: n)
hipError_t transpose64(hipStream_t s, int n, const double* d_src, double* d_dst, int lds = 0, int ldd = 0);
hipError_t symm_lower64(hipStream_t s, int n, const double* d_src, double* d_dst);  // mirror the lower triangle
hipError_t fill64(hipStream_t s, double* d_p, size_t n, double v);
// d_dst (n x n) = diag(dl) src[:n, :n] diag(dr); src column stride lds
hipError_t scale_rc64(hipStream_t s, int n, const double* d_src, int lds, const double* d_dl, const double* d_dr, double* d_dst);

// ---- level-sorted rows (sorted.hip): the pixel halves of the table passes without LDS atomics
constexpr int kSortedThreads = 512;
struct SortedRows {
    const unsigned short* scol;   // [nrows][pitch] per row: one slot of 8 x column per chunk (sample pixels left out)
    const uint2* desc;            // [nrows][kSortedThreads] chunk of each pass thread
    const unsigned short* first;  // [nrows][258] first chunk of each level; [257] tree steps
    const double* E;              // [W + 1] exp(-d^2 / hx^2)
    bool rec;                     // column factors by recurrence from two table reads (sorted_recurrence)
    double kappa;                 // exp(-2 colStep^2 / hx^2)
    const double* E2 = nullptr;   // [W + 1] exp(-2 d^2 / hx^2): set (with hx) where the Gram runs on index sums (sorted_gsum_ok)
    double hx = 0.0;
    int lev_t0 = 0, lev_nt = 16;  // the 16-level tiles [lev_t0, lev_t0 + lev_nt) that occur in the image (check_levels)
    bool mom = false;             // the pass kernel's pixel loop in its moment form (sorted_moments_ok)
};
// Gram by index sums (sorted.hip: k_sorted_gsum): S_r[t][x] = sum c_i^2 G_t(col_i), t < 2 nC - 1; layout [row][t][level]
bool sorted_gsum_ok(GridSpec gs, double hx);
hipError_t sorted_gram_sums(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                            const unsigned short* d_first, const double* d_E2, const double* d_cvec, double* d_Aout, double hx);
bool sorted_recurrence(GridSpec gs, double hx, double* kappa);
int sorted_max_width();
size_t sorted_scol_elems(int W, int nrows_local);  // allocation size of SortedRows::scol
int sorted_gram_max_cols();
// expand half of the sample-space apply on the sorted rows (nl <= sorted_expand_layers() layers per launch)
int sorted_expand_max_cols();
int sorted_expand_max_width();
int sorted_expand_layers(GridSpec gs);
hipError_t sorted_expand(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                         const double* d_E, const double* d_g, size_t gstride, int nl, const double* d_cvec, float* d_out,
                         long long ostride, bool rec = false, double kappa = 0.0, bool round8 = false);
hipError_t dist_table(hipStream_t s, int W, double hx, double* d_E);
hipError_t sort_rows(hipStream_t s, const float* d_lum, GridSpec gs, int row0, int nrows_local, unsigned short* d_scol,
                     uint2* d_desc, unsigned short* d_first);
hipError_t sorted_pass(hipStream_t s, int mode, GridSpec gs, int row0, int nrows_local, const unsigned short* d_scol,
                       const uint2* d_desc, const unsigned short* d_first, const double* d_E, const double* d_g, double eps,
                       double* d_ybuf, double* d_h, const double* d_cvec, const float* d_xvec, bool rec, double kappa,
                       int lev_t0 = 0, int lev_nt = 16,   // table columns of the level tiles [lev_t0, +lev_nt) only
                       bool mom = false);                 // the moment form of the pixel loop (sorted_moments_ok)
bool sorted_moments_ok(GridSpec gs, double hx);
hipError_t sorted_gram_rows(hipStream_t s, GridSpec gs, int nrows_local, const unsigned short* d_scol, const uint2* d_desc,
                            const unsigned short* d_first, const double* d_E, const double* d_cvec, double* d_Aout, bool rec,
                            double kappa);

// tiled form of sink_hist (three kernels, Ep read once per pass); writes the full column sums to d_z
size_t hist_tiled_workspace_elems(GridSpec gs, int nrows_local);
hipError_t sink_hist_tiled(hipStream_t s, int mode, const float* d_lum, GridSpec gs, int p, int ldp, int row0,
                           int nrows_local, const double* d_er, const double* d_ecT, const double* d_Ep,
                           const double* d_w, double eps, double* d_ybuf, double* d_ws, double* d_z,
                           LaunchObserver* obs = nullptr, const double* d_cvec = nullptr,
                           const float* d_xvec = nullptr,   // mode XVEC: y_i = cvec_i * xvec_i (apply, reduce half)
                           const SortedRows* sorted = nullptr);  // given: the pixel kernel runs on the level-sorted rows
// sample-space apply (tables): expand half for one layer, the p/K-sized middle, and the sample-pixel outputs
int apply_layers_per_launch(GridSpec gs);
// (sorted given, nC <= 12, W <= 4096: the pixel kernel runs on the level-sorted rows, sorted_expand)
hipError_t apply_hist_layers(hipStream_t s, const float* d_lum, GridSpec gs, int p, int row0, int nrows_local,
                             const double* d_er, const double* d_ecT, const double* d_Ep, const double* d_wl, int ldw,
                             int nl, const double* d_c, double* d_ws, float* d_out, long long ostride,
                             LaunchObserver* obs, const SortedRows* sorted = nullptr, bool round8 = false);
hipError_t apply_small(hipStream_t s, int p, int K, int ldk, int L, int ldw, const double* d_m, const double* d_D,
                       const double* d_Vrows, const double* d_xA /* x at the p sample pixels */, const double* d_resp,
                       double* d_t, double* d_Wp, double* d_YA);
hipError_t scatter_samples(hipStream_t s, int p, int L, const long long* d_loc, const double* d_YA, float* d_Y,
                           long long ystride, bool round8 = false);  // round8: clamp to [0, 255], round half to even, in fp64

// Gram in sample space through the same tables (quantised luminance, nSelCols <= ghist_max_cols())
int ghist_max_cols();
size_t ghist_workspace_elems(GridSpec gs, int nrows_local);
hipError_t gram_hist(hipStream_t s, const float* d_lum, GridSpec gs, int p, int row0, int nrows_local,
                     const double* d_er, const double* d_ecT, const double* d_Ep, const double* d_c, double* d_ws,
                     double* d_Gk, LaunchObserver* obs = nullptr, const SortedRows* sorted = nullptr);

// projection through the tables (quantised luminance): V = diag(c) K D, one workgroup per image row
bool project_hist_ok(GridSpec gs, int p, int K);
hipError_t project_hist(hipStream_t s, const float* d_lum, GridSpec gs, int p, int row0, int nrows_local,
                        const double* d_er, const double* d_ecT, const double* d_Ep, const double* d_D, int ldd, int K,
                        const double* d_c, float* d_V, int ldv);

// 8-bit BGR <-> Lab (colour.hip): d_lut = the fixed-point tables of nle_lab8_tables as one blob; d_lab / d_L optional outputs
hipError_t bgr2lab8(hipStream_t s, const unsigned char* d_bgr, long long n, const double* d_lut, unsigned char* d_lab,
                    float* d_L);
hipError_t lab2bgr8(hipStream_t s, const unsigned char* d_lab, const float* d_L, const float* d_a, const float* d_b,
                    long long n, const double* d_lut, unsigned char* d_bgr);
hipError_t channel8(hipStream_t s, const unsigned char* d_img, long long n, int ch, float* d_out);
// cv::max(y, 0) / cv::min(y, 255) / convertTo(CV_8U) of a filtered plane (src/filter.cpp:434-436): round half to even
hipError_t plane_to_u8(hipStream_t s, const float* d_y, long long n, unsigned char* d_out);
hipError_t channel8_plane(hipStream_t s, const unsigned char* d_u8, long long n, float* d_out);  // bytes -> fp32 levels
// single-channel 8-bit bilateral filter (fp32 planes holding integers); tables from the host: space_w (2r+1)^2 with 0
// outside the circle, colour_w 256 entries
int bilateral8_max_radius();
hipError_t bilateral8(hipStream_t s, const float* d_src, int H, int W, int radius, const float* d_space_w,
                      const float* d_colour_w, float* d_dst);

// out[block][2*ncols] = {min, max} of the first ncols columns of X over the block's rows
hipError_t col_range(hipStream_t s, const float* d_X, long long M, int ld, int ncols, float* d_out, int nblocks);

// Y[l][i] = sum_k V[i][k] * g[l][k]   (g: L x ld doubles, device)
hipError_t apply_expand(hipStream_t s, const float* d_V, long long M, int ld, const double* d_g,
                        int L, float* d_Y, long long ystride);

// X[idx[k]] = src[k] (ld floats each) for idx[k] in [0, M)
hipError_t scatter_rows(hipStream_t s, const float* d_src, const long long* d_idx, int n, int ld,
                        float* d_X, long long M);

}  // namespace nlek
