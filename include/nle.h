/*
 * nle.h -- C ABI of the MI355X-native nonlocal-filter hot path.
 *
 * This is the drop-in boundary for the path `src/filter.cpp` of
 * lightalchemist/nonlocal-image-edit names (computeKernel -> nystromApproximation ->
 * sinkhorn -> orthogonalize -> P*V, then NLEFilter::apply).  The reference has no FFI
 * of its own (it is one C++ translation unit calling Eigen/OpenCV); the symbols below
 * are what a binding for this path binds, and `include/nle/filter.hpp` is the C++
 * surface with the reference's own names layered over them.  Each entry point cites
 * the reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 == NLE_OK);
 *     the message for the last failure is `nle_last_error(ctx)`.
 *   - "d_" pointers are DEVICE (HIP) pointers, "h_" pointers are HOST pointers.
 *   - N-sized matrices are fp32, ROW-PER-PIXEL (row-major, leading dimension `ld`,
 *     a multiple of 4; columns >= the logical width are zero), in NATURAL pixel order
 *     (row-major image scan), not the reference's [selected; rest] order.
 *   - small (p x p, r x r, r- and K-sized) quantities are fp64 on the host, matrices
 *     COLUMN-MAJOR like the reference's Eigen::MatrixXd (include/filter.hpp:10-12).
 *   - all device work is stream-ordered on the ctx's stream; a ctx is used by one
 *     host thread at a time (the reference is single-threaded and keeps no globals).
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails
 *     with NLE_ERR_HIP.
 */
#ifndef NLE_H
#define NLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLE_OK 0
#define NLE_ERR_INVALID 1  /* bad argument / shape; message mirrors the reference's runtime_error text */
#define NLE_ERR_HIP 2      /* HIP runtime failure (incl. no device) */
#define NLE_ERR_NUMERIC 3  /* eigensolver did not converge / empty spectrum */
#define NLE_ERR_COMM 4     /* all-reduce callback failed */

#define NLE_EPS 1e-10 /* include/filter.hpp:14 */

typedef struct nle_ctx nle_ctx;
typedef struct nle_filter nle_filter;

/* ---- context ------------------------------------------------------------------- */
/* `stream` is a hipStream_t to run on, or NULL to let the ctx create and own a non-blocking
 * stream.  NB: the handle of HIP's legacy default stream IS NULL, so it cannot be passed; a host
 * that wants its own collectives ordered with the ctx's kernels (torch.distributed) creates a
 * stream, hands its handle over and issues the collectives on it (the Python mirror does). */
int nle_ctx_create(int device, void* stream, nle_ctx** out);
void nle_ctx_destroy(nle_ctx* ctx);
const char* nle_last_error(const nle_ctx* ctx); /* ctx may be NULL: last create error */
int nle_ctx_synchronize(nle_ctx* ctx);
/* The ctx keeps the device workspace of finished calls (and of destroyed filters) for reuse by
 * the next call; nle_ctx_trim returns it to the driver (nle_ctx_destroy does so too). */
int nle_ctx_trim(nle_ctx* ctx);

/* Device memory helpers for hosts that do not link the HIP runtime themselves (the C++ surface in
 * include/nle/filter.hpp, a cgo/JNI/ctypes binding): plain hipMalloc / hipFree / hipMemcpy on
 * the ctx's device, synchronous with respect to the ctx's stream. */
int nle_dev_alloc(nle_ctx* ctx, size_t bytes, void** d_ptr);
void nle_dev_free(nle_ctx* ctx, void* d_ptr);
/* page-locked host memory (hipHostMalloc): host buffers handed to the *_host entry points are copied at PCIe speed
 * and asynchronously only when they are pinned */
int nle_host_alloc(nle_ctx* ctx, size_t bytes, void** h_ptr);
void nle_host_free(nle_ctx* ctx, void* h_ptr);
int nle_dev_upload(nle_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int nle_dev_download(nle_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* Which formulation nle_train uses for the N-sized passes:
 *   NLE_MODE_AUTO          the table form of NLE_MODE_PHI_FREE when it applies (integer-valued luminance plane -- what the
 *                          reference always feeds, the L channel of 8-bit Lab, src/filter.cpp:460-469 -- a sample grid of at
 *                          most 32 x 36, <= 128 eigenvectors), else NLE_MODE_MATERIALISED_F64.  Both are fp64 from the
 *                          affinities to the last reduction and meet the 1e-4 per-layer bar wherever the reference
 *                          algorithm itself is well posed; nle_filter_diag reports which one ran.  The two fp32
 *                          formulations below run only when asked for: they are the north star's literal kernels
 *                          (fp32 MFMA GEMMs, HBM-streamed Sinkhorn) and can miss the bar on inputs whose detail layers
 *                          are small differences (DESIGN.md "Numerics").
 *   NLE_MODE_MATERIALISED  Phi = K_AB^T V_A Lambda^-1 is written once (N x r fp32) and streamed
 *   NLE_MODE_PHI_FREE      every pass regenerates its affinity rows in registers; when the luminance
 *                          plane is integer valued in [0, 255] (the L channel of 8-bit Lab, what the
 *                          reference always feeds, src/filter.cpp:460-469) the Sinkhorn passes use
 *                          fp64 look-up tables instead of exponentials
 *   NLE_MODE_PHI_FREE_EXP  Phi-free without the look-up-table specialisation */
#define NLE_MODE_AUTO 0
#define NLE_MODE_MATERIALISED 1
#define NLE_MODE_PHI_FREE 2
#define NLE_MODE_PHI_FREE_EXP 3
/*   NLE_MODE_MATERIALISED_F64  the literal decomposition with Phi (N x r) and V (N x K') in fp64: fp64 affinities,
 *                          fp64-MFMA products, any luminance, up to 2048 samples (Phi must fit in device memory) */
#define NLE_MODE_MATERIALISED_F64 4
/*   NLE_MODE_STREAMED_F64  fp64 like NLE_MODE_MATERIALISED_F64 but without the N x r matrix: the sample-space algebra of
 *                          NLE_MODE_PHI_FREE on fp64 affinity rows regenerated chunk by chunk (libm exp) in every pass; the
 *                          workspace is bounded (NLE_STREAM64_CHUNK_MB, default 2048), V (N x K') is fp64.  Any plane, any
 *                          grid with <= 2048 samples, any K: what auto mode takes when Phi would not fit the device. */
#define NLE_MODE_STREAMED_F64 5
int nle_ctx_set_mode(nle_ctx* ctx, int mode);
/* The Nystrom-extension GEMM Phi = K_AB^T (V_A Lambda^-1) (src/filter.cpp:275) of NLE_MODE_MATERIALISED and of nle_nystrom on
 * the bf16 matrix cores with SPLIT operands (each fp32 value as three bf16, six products, fp32 accumulate: fp32 accuracy at
 * ~2.7x the exact-fp32 MFMA's rate; plain bf16 operands miss the parity bar).  Off by default. */
int nle_ctx_set_nystrom_bf16x3(nle_ctx* ctx, int on);

/* Slab input (multi-GPU, SURVEY.md section 8e "each GPU uploads / downloads only its slab"): with on != 0 every plane
 * handed to nle_train*, nle_apply* on this ctx holds ONLY the rows [row0, row1) of this rank (nle_slab_rows), n_local
 * values, instead of the full H x W image; H and W stay the full image's.  The p sample values (and x at the sample
 * pixels in apply) are then completed across the ranks by one small all-reduce each.  Set it after the shard /
 * communicator; it has no effect at world == 1. */
int nle_ctx_set_slab_input(nle_ctx* ctx, int on);

/* How orthogonalize finds the top eigenpairs of Q (src/filter.cpp:310-317).  0 (default): the reference's default build,
 * eigenDecomposition(Q) -- a full eigensolve, the first min(nEigenVectors, kept) pairs.  1 (opt-in; also
 * NLE_Q_SOLVER=lanczos in the environment): the reference's USE_SPECTRA build, topkEigenDecomposition (:170-199) --
 * Lanczos for the nev = min(nEigenVectors, q - 1) pairs of largest magnitude, Krylov dimension min(2 nev, q), tolerance
 * 1e-10, at most 1000 restarts, the FULL (not triangle-mirrored) Q, converged pairs only.  Results agree with the
 * full solver to the tolerance; K' is capped at q - 1. */
int nle_ctx_set_topk_solver(nle_ctx* ctx, int solver);

/* Multi-GPU (one process per GPU).  Rank `rank` of `world` owns image rows
 * [rank*H/world, (rank+1)*H/world).  `allreduce(user, d_buf, count)` must sum, in place
 * and stream-ordered with the ctx's stream, `count` doubles at DEVICE pointer d_buf over
 * all ranks (torch.distributed / RCCL all_reduce on that stream).  `d_comm` is a device
 * buffer of `comm_len` doubles owned by the caller, comm_len >= nle_comm_len(p).
 * world == 1 (default) needs no callback.  New in this build (the reference is
 * single-process). */
/* Native form (preferred): the library calls RCCL itself -- ncclAllReduce, fp64 sum, IN PLACE on the ctx's stream, no
 * staging copies and no host callback.  librccl.so is loaded on first use.  One process per GPU: rank 0 obtains an id
 * with nle_rccl_unique_id and hands the 128 bytes to the other ranks by whatever channel the launcher has (an MPI /
 * torch.distributed broadcast, a file); every rank then calls nle_ctx_init_rccl (collective: ncclCommInitRank).  A host
 * that already owns a communicator for the ctx's device passes it to nle_ctx_set_rccl_comm instead (borrowed). */
#define NLE_RCCL_UNIQUE_ID_BYTES 128
int nle_rccl_unique_id(void* h_id, size_t size);
int nle_ctx_init_rccl(nle_ctx* ctx, int rank, int world, const void* h_id, size_t size);
int nle_ctx_set_rccl_comm(nle_ctx* ctx, int rank, int world, void* nccl_comm);
/* Error path of a multi-rank host: aborts the ctx's own communicator (ncclCommAbort) so that collectives of this ctx that
 * are waiting for a rank that has failed end instead of blocking for ever; callable from another thread than the one using
 * the ctx.  Afterwards every call on the ctx that needs a collective returns NLE_ERR_COMM until nle_ctx_init_rccl binds a
 * new communicator.  No-op without an own communicator. */
int nle_ctx_abort_rccl(nle_ctx* ctx);
/* Callback form (any transport, e.g. gloo in the CPU rehearsal tests): */
typedef int (*nle_allreduce_fn)(void* user, void* d_buf, size_t count);
int nle_ctx_set_shard(nle_ctx* ctx, int rank, int world, nle_allreduce_fn allreduce,
                      void* user, double* d_comm, size_t comm_len);
size_t nle_comm_len(int n_samples);

/* ---- host-only helpers (no GPU needed) ----------------------------------------- */
/* samplePixels, src/filter.cpp:56-80, in closed form: the selected set is
 * {row_off + i*row_step, i < n_sel_rows} x {col_off + j*col_step, j < n_sel_cols}. */
int nle_sample_grid(int H, int W, int n_row_samples, int n_col_samples, int* row_step,
                    int* row_off, int* n_sel_rows, int* col_step, int* col_off,
                    int* n_sel_cols);
/* rows [*row0, *row1) owned by `rank` of `world` */
int nle_slab_rows(int H, int rank, int world, int* row0, int* row1);
/* eigenDecomposition, src/filter.cpp:204-228 (decl include/filter.hpp:23-24): symmetric
 * n x n (col-major, LOWER triangle read), eigenvalues DESCENDING, leading run >= eps kept.
 * h_U: n x n col-major (first *r columns valid), h_D: n (first *r valid). */
int nle_eigen_decomposition(const double* h_M, int n, double eps, double* h_U, double* h_D,
                            int* r);
/* the same when only the first kmax eigenvectors are wanted (orthogonalize keeps nEigVectors columns of Q's, :313-316):
 * h_D receives ALL n eigenvalues (descending), h_U (n x min(kmax, n)) the first min(kmax, n) eigenvectors, *r the length
 * of the leading run >= eps.  For kmax <= n / 2 the eigenvectors come from inverse iteration on the tridiagonal form. */
int nle_eigen_decomposition_top(const double* h_M, int n, double eps, int kmax, double* h_U, double* h_D, int* r);
/* the form the train path itself uses for Q on the host: h_Dk receives only the min(kmax, n) LARGEST eigenvalues (descending),
 * h_U (n x min(kmax, n)) their eigenvectors, *r the number of eigenvalues >= eps -- all orthogonalize reads of Q's
 * decomposition (:313-316).  For 2 kmax <= n the eigenvalues come from bisection on Sturm counts of the tridiagonal form
 * (the count at eps is *r) instead of a QL run for all n; otherwise this is nle_eigen_decomposition_top. */
int nle_eigen_decomposition_topk(const double* h_M, int n, double eps, int kmax, double* h_U, double* h_Dk, int* r);
/* the same with the Householder reduction to tridiagonal form on the GPU (one workgroup, the matrix in registers; n <= 224)
 * and the O(n^2) rest on the host -- what nle_train* uses for Q when K <= q / 2 and q <= 224 (NLE_HOST_TRIDIAG=1 keeps
 * the whole solve on the host).  Same conventions; results agree with the host form to rounding. */
int nle_eigen_decomposition_top_device(nle_ctx* ctx, const double* h_M, int n, double eps, int kmax, double* h_U, double* h_D,
                                       int* r);
/* eigenDecomposition (:204-228) with the O(n^3) part on the GPU, 3 <= n <= 1152: Householder reduction in one persistent
 * multi-workgroup launch, all eigenvalues by Sturm bisection, the eigenvectors of h_D[first .. first + count) by inverse
 * iteration on the tridiagonal form (host) and their back-transformation (device).  h_D: all n eigenvalues descending;
 * h_U: n x count col-major (may be NULL when count == 0); *r = length of the leading run >= eps.  What nle_train* uses
 * for Wa and Q from 288 samples on (NLE_DEV_SOLVER_MIN; NLE_HOST_SOLVER=1 keeps the host solver). */
int nle_sym_eigen_device(nle_ctx* ctx, const double* h_M, int n, double eps, int first, int count, double* h_U, double* h_D,
                         int* r);
/* Cholesky factor of a symmetric matrix (lower triangle read) and its inverse on the GPU (blocked, fp64 MFMA updates):
 * h_L, h_Linv n x n col-major lower triangular, *inv_trace = trace(M^-1), *ok = 0 if a pivot was not positive. */
int nle_cholesky_device(nle_ctx* ctx, const double* h_M, int n, double* h_L, double* h_Linv, double* inv_trace, int* ok);
/* topkEigenDecomposition, src/filter.cpp:170-199 (the USE_SPECTRA build's solver for Q): the min(n_largest, n - 1)
 * eigenpairs of largest magnitude of the FULL n x n matrix by Lanczos (tolerance 1e-10, <= 1000 restarts), algebraic
 * value descending, leading run >= eps kept.  h_U: n x min(n_largest, n - 1) col-major, h_D likewise; *r valid pairs. */
int nle_topk_eigen_decomposition(const double* h_M, int n, int n_largest, double eps, double* h_U, double* h_D, int* r);
/* transformEigenValues, src/filter.cpp:334-347 */
int nle_transform_eigenvalues(const double* h_eigvals, int K, const double* h_weights, int L,
                              double* h_fS);
/* per-layer spectral responses implied by :334-347: h_resp[l*K + k] */
int nle_layer_responses(const double* h_eigvals, int K, int L, double* h_resp);

/* ---- stage-level entry points (unit parity; all sizes generic) ------------------- */
/* computeKernel, src/filter.cpp:114-167 (decl include/filter.hpp:20-21), for this rank's
 * slab.  d_lum: FULL H x W luminance plane (fp32).  Outputs: h_Ka (p x p col-major fp64,
 * may be NULL), d_kab: n_local x ld fp32 affinity rows k_i[s] = exp(negDist(pixel i,
 * sample s)) for EVERY local pixel in natural order (sample pixels included: their rows
 * are rows of Ka).  ld = nle_ld(p).  The permutation `P` (:156-164) is implicit:
 * nle_sample_grid gives it in closed form. */
int nle_compute_kernel(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples,
                       int n_col_samples, double hx, double hy, double* h_Ka, float* d_kab);
/* nystromApproximation, src/filter.cpp:257-280 (decl include/filter.hpp:26-27), fused with
 * the affinity evaluation (K_AB is never written): h_eigvals[r], d_phi n_local x nle_ld(r)
 * (allocated by the caller for r = p, i.e. n_local * nle_ld(p) floats). */
int nle_nystrom(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples,
                int n_col_samples, double hx, double hy, double* h_eigvals, int* r,
                float* d_phi);
/* Generic tall-skinny product of :275 for caller-supplied matrices:
 * d_C (M x nle_ld(nc)) = d_A (M x lda, logical width kd) * h_B (kd x nc col-major fp64). */
int nle_ts_gemm(nle_ctx* ctx, const float* d_A, long long M, int lda, int kd,
                const double* h_B, int nc, float* d_C);
/* sinkhorn iterations, src/filter.cpp:238-245 (decl include/filter.hpp:29-30), on a
 * device-resident phi (M x ld, logical width r): returns the two r-vectors
 * u_c = lambda o Phi^T r_{T-1}, u_r = lambda o Phi^T c_T that define the final scalings
 * c_i = recip(phi_i . u_c), r_i = recip(phi_i . u_r).  max_iter >= 1. */
int nle_sinkhorn_scalings(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r,
                          const double* h_eigvals, int max_iter, double* h_u_c,
                          double* h_u_r);
/* Gram matrix of the scaled rows, the N-sized half of `Wab*Wab^T` (src/filter.cpp:296):
 * h_G (r x r col-major) = sum_i c_i^2 phi_i phi_i^T, c_i = recip(phi_i . h_u) (all rows);
 * h_u == NULL: c_i = 1 (plain X^T X, used for `Wab*Wab^T` on a caller-supplied Wab). */
int nle_gram(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r, const double* h_u,
             double* h_G);
/* per-row scalings c_i = recip(phi_i . h_u) (inplaceReciprocal, src/filter.cpp:42-54) */
int nle_row_scalings(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r,
                     const double* h_u, double* d_out);

/* The same stages on fp64 device matrices (row-per-pixel, leading dimension any value >= the logical width; outputs use
 * nle_ld(width)), all products and sums in fp64: what include/nle/filter.hpp's free functions run on, so that the
 * reference's unit tests (test/test_filter.cpp, tolerance 1e-10) hold at their own tolerance. */
int nle_compute_kernel64(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                         double hy, double* h_Ka, double* d_kab);
int nle_ts_gemm64(nle_ctx* ctx, const double* d_A, long long M, int lda, int kd, const double* h_B, int nc, double* d_C);
int nle_sinkhorn_scalings64(nle_ctx* ctx, const double* d_phi, long long M, int ld, int r, const double* h_eigvals,
                            int max_iter, double* h_u_c, double* h_u_r);
int nle_gram64(nle_ctx* ctx, const double* d_phi, long long M, int ld, int r, const double* h_u, double* h_G);
int nle_row_scalings64(nle_ctx* ctx, const double* d_phi, long long M, int ld, int r, const double* h_u, double* d_out);

/* ---- the fused path ---------------------------------------------------------------- */
/* NLEFilter::trainFilter, src/filter.cpp:480-502.  d_lum: FULL H x W fp32 luminance on the
 * device.  The filter keeps V (n_local x ld(K') fp32, pixel order) and eigvals on the
 * device/host: the reference's m_eigvecs / m_eigvals (include/filter.hpp:52-53).
 * Errors: n_row_samples > H or n_col_samples > W -> NLE_ERR_INVALID with the reference's
 * message (:117-119). */
int nle_train(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples,
              int n_col_samples, double hx, double hy, int n_sinkhorn_iter,
              int n_eigen_vectors, nle_filter** out);
/* same, luminance plane in HOST memory (what NLEFilter::trainFilter is handed) */
int nle_train_host(nle_ctx* ctx, const float* h_lum, int H, int W, int n_row_samples,
                   int n_col_samples, double hx, double hy, int n_sinkhorn_iter,
                   int n_eigen_vectors, nle_filter** out);
/* same, from the 8-bit plane itself: what getLuminanceChannel (src/filter.cpp:460-469) hands trainFilter is the L channel of
 * an 8-bit Lab image converted to double -- h_lum8 holds those bytes (H x W, this rank's rows in slab-input mode), one byte
 * per pixel crosses PCIe instead of four and the device widens them.  The filter is the one nle_train_host builds from the
 * same levels as floats (bit-identical), and it keeps the plane likewise for nle_apply*_host(h_x == NULL). */
int nle_train_host_u8(nle_ctx* ctx, const unsigned char* h_lum8, int H, int W, int n_row_samples,
                      int n_col_samples, double hx, double hy, int n_sinkhorn_iter,
                      int n_eigen_vectors, nle_filter** out);
void nle_filter_destroy(nle_filter* f);
/* any out pointer may be NULL.  n_local = pixels of this rank's slab, K = kept eigenpairs
 * (K' of src/filter.cpp:314), r = retained rank of Ka, p = realised sample count. */
int nle_filter_info(const nle_filter* f, long long* n_local, int* K, int* r, int* p,
                    int* row0, int* row1);
/* What the last train decided, h_info[8]:
 *   [0] formulation of the N-sized passes that was taken (NLE_MODE_MATERIALISED, NLE_MODE_PHI_FREE = look-up
 *       tables, NLE_MODE_PHI_FREE_EXP) -- auto mode resolves to one of these per image;
 *   [1] p realised sample count;  [2] eigenvalues of Ka kept by the cut at 1e-10 (src/filter.cpp:214,262-271);
 *   [3] eigenvalues of Wa kept (:287);  [4] eigenvalues of Q kept (:313);  [5] K' = min(nEigenVectors, [4]) (:314);
 *   [6] 1 if Ka was factored by Cholesky (certified full rank at the cut), 0 if by the eigensolver;
 *   [7] the same for Wa's inverse root. */
int nle_filter_diag(const nle_filter* f, int* h_info);
int nle_filter_eigvals(const nle_filter* f, double* h_eigvals /* K */);
/* min / max coefficient of the first `ncols` eigenvectors over this rank's slab (what the reference
 * prints at src/filter.cpp:506): h_min[ncols], h_max[ncols].  A filter whose V is implicit projects just
 * these columns into a temporary; it does not materialise the N x K matrix. */
int nle_filter_eigvec_range(const nle_filter* f, int ncols, double* h_min, double* h_max);
/* device pointer + leading dimension of V (n_local x ld), for inspection.  In the table formulation
 * the filter keeps V implicit (V = diag(c) K D) and applies it on its p-sized side; the first call of
 * this / nle_filter_copy_eigvecs / nle_filter_eigvec_range materialises the matrix (one projection GEMM). */
int nle_filter_eigvecs(const nle_filter* f, const float** d_V, int* ld);
/* copy V (n_local x ld floats) into a caller-owned DEVICE buffer */
int nle_filter_copy_eigvecs(const nle_filter* f, float* d_out);
/* per-stage milliseconds of the last train: sample fetch + Ka eigensolve, sinkhorn (incl. building
 * Phi in the materialised mode), gram, project (HIP events); host algebra; wall total.  h_ms[6] */
int nle_filter_timings(const nle_filter* f, double* h_ms);

/* NLEFilter::apply, src/filter.cpp:445-458: y = V diag(fS) V^T x for this rank's slab.
 * d_x: FULL H x W fp32 channel on the device; d_y: n_local fp32.  Size mismatch with the
 * training image -> NLE_ERR_INVALID (:447-449). */
int nle_apply(nle_filter* f, const float* d_x, int H, int W, const double* h_fS, float* d_y);
/* the L per-layer outputs y_l = V ((lambda^l - lambda^(l+1)) o V^T x), base layer
 * lambda^(L-1) (src/filter.cpp:334-347); d_y: L x n_local fp32 (layer-major). */
int nle_apply_layers(nle_filter* f, const float* d_x, int H, int W, int L, float* d_y);
/* host-buffer forms.  h_x == NULL: filter the plane the filter was trained on (kept on the device by
 * nle_train_host), which is what `enhance` does (src/enhance.cpp:43-44) -- one upload for train + apply.  Finished
 * layers are copied back on a second stream while the next one is computed. */
int nle_apply_host(nle_filter* f, const float* h_x, int H, int W, const double* h_fS,
                   float* h_y);
int nle_apply_layers_host(nle_filter* f, const float* h_x, int H, int W, int L, float* h_y);
/* The L plane NLEFilter::enhance merges back (src/filter.cpp:428-436): y = apply(x, fS), cv::max(y, 0), cv::min(y, 255),
 * convertTo(CV_8U) (round half to even, saturate) -- n_local BYTES instead of L fp32 planes: what `enhance` needs back from
 * the device (16 MB instead of 268 MB at 4096^2, four weights).  Device and host-buffer forms; h_x == NULL as above. */
int nle_apply_u8(nle_filter* f, const float* d_x, int H, int W, const double* h_fS, unsigned char* d_out);
/* the same plane as fp32 holding the 8-bit levels (the replacement-channel argument of nle_lab2bgr8 / _planes).  On the
 * default path the clamp and the round-half-even act on the fp64 value of the filtered plane, before anything is rounded
 * to fp32: `enhance` then rounds as the reference's fp64 pipeline does, not an fp32 plane's ties. */
int nle_apply_rounded8(nle_filter* f, const float* d_x, int H, int W, const double* h_fS, float* d_y);
int nle_apply_u8_host(nle_filter* f, const float* h_x, int H, int W, const double* h_fS, unsigned char* h_out);

/* ---- colour wrapper on the device (the code either side of the path) ------------------------------- */
/* cv::cvtColor(COLOR_BGR2Lab) on an 8-bit image as the reference uses it (src/filter.cpp:423,463) and
 * the split / convertTo(CV_64F) of the L channel (:424-426,465-467): d_bgr n x 3 bytes -> d_lab n x 3
 * bytes (may be NULL) and d_L n floats = L in 0..255 (may be NULL).  OpenCV's 8-bit path is not the float formula of its
 * documentation but a fixed-point table algorithm (imgproc `RGB2Lab_b`); that algorithm is what runs here -- with it the
 * reference's README outputs are reproduced to 0.003 .. 0.5 grey levels (tests/test_oracle_readme_pairs.py). */
int nle_bgr2lab8(nle_ctx* ctx, const unsigned char* d_bgr, long long n, unsigned char* d_lab, float* d_L);
/* the tables of that conversion (host only): h_gamma[256] = sRGB decode scaled by 255 * 8, h_cbrt[3072] = f(t) of L*a*b*
 * scaled by 2^15 (t = i / 2040; made in single precision with OpenCV's truncating rational cube root, csrc/lab8_tables.cpp),
 * h_coeffs[9] = the XYZ matrix over the D65 white point in 12-bit fixed point, row major (R, G, B columns).  L = (296 fY - 1336934 + 2^14) >> 15, a = (500 (fX - fY) + 128 * 2^15 + 2^14) >> 15, b alike with
 * 200 (fY - fZ); f* = h_cbrt[(R c0 + G c1 + B c2 + 2^11) >> 12]. */
int nle_lab8_tables(unsigned short* h_gamma, unsigned short* h_cbrt, int* h_coeffs);
/* the tables of the inverse conversion (host only), everything in units of 2^-14: h_yf[256][2] = (y, fy) for an 8-bit L,
 * h_ab_to_xz[36864] = the inverse of f(t) for t = index - 8145 (t <= 3390 ? t 108 / 841 - 290 : t t / 2^14 t / 2^14, C
 * integer division), h_inv_gamma[4096] = round(255 sRGB_encode(i / 4096)), h_coeffs[9] = round(2^12 XYZ -> sRGB times the
 * D65 white point), rows R, G, B.  fx = fy + ((5 a 53687 + 2^7) >> 13) - 4194, fz = fy - (((b 41943 + 2^4) >> 9) - 10484);
 * channel = h_inv_gamma[clamp((c0 x + c1 y + c2 z + 2^13) >> 14, 0, 4095)]. */
int nle_lab8_inverse_tables(unsigned short* h_yf, int* h_ab_to_xz, unsigned short* h_inv_gamma, int* h_coeffs);
/* max(0) / min(255) / convertTo(CV_8U) / merge / cvtColor(COLOR_Lab2BGR) (src/filter.cpp:434-440): the L
 * channel is taken from d_L (clamped, rounded half to even) when given, else from d_lab.  The conversion is OpenCV's
 * integer 8-bit path (imgproc `Lab2RGBinteger`, tables below): with it `enhance` writes the author's README output files
 * byte for byte wherever the filtered L plane has no rounding tie (tests/test_readme_pairs_gpu.py). */
int nle_lab2bgr8(nle_ctx* ctx, const unsigned char* d_lab, const float* d_L, long long n, unsigned char* d_bgr);
/* the same with any of the three channels replaced by an fp32 plane (NULL = keep d_lab's), each clamped and
 * rounded like :391-399 -- the tail of NLEFilter::denoise (src/filter.cpp:391-409) */
int nle_lab2bgr8_planes(nle_ctx* ctx, const unsigned char* d_lab, const float* d_L, const float* d_a, const float* d_b,
                        long long n, unsigned char* d_bgr);
/* cv::split + convertTo of one channel (0, 1, 2) of an interleaved 8-bit 3-channel image (:363,376-378) */
int nle_lab8_channel(nle_ctx* ctx, const unsigned char* d_lab, long long n, int channel, float* d_out);

/* ---- denoise wrapper (src/denoise.cpp, src/filter.cpp:349-410,521-538): the bilateral prefilter ------------- */
/* cv::bilateralFilter(src, dst, -1, sigmaColor, sigmaSpace, BORDER_DEFAULT) on a single-channel 8-bit plane, as
 * OpenCV documents it for CV_8UC1: radius = round(1.5 sigmaSpace) >= 1, circular window, fp32 weights
 * space[dy,dx] * colour[|v - v0|], fp32 sums in row-major window order, round-half-even of sum / wsum,
 * BORDER_REFLECT_101.  Planes are fp32 holding integers 0..255 (what nle_bgr2lab8 / nle_lab8_channel produce and
 * nle_train consumes); d_dst must not alias d_src.  OpenCV's own build (SIMD summation order, version) is not
 * available here, so agreement with it is unpinned beyond the documented formula. */
int nle_bilateral8(nle_ctx* ctx, const float* d_src, int H, int W, double sigma_color, double sigma_space, float* d_dst);
/* the two weight tables of that filter (host): *radius always; h_space_w (2 radius + 1)^2 floats with 0 outside
 * the circle and h_colour_w 256 floats when both are non-NULL */
int nle_bilateral_tables(double sigma_color, double sigma_space, int* radius, float* h_space_w, float* h_colour_w);

/* leading dimension used for a logical width n: (n + 3) & ~3 */
int nle_ld(int n);

/* ---- measurement hooks (bench.py) --------------------------------------------------- */
/* Kernel ids for the per-kernel HIP-event timing below. */
enum {
    NLE_K_AFFINITY = 0,      /* materialising affinity pass (stage API: computeKernel)   */
    NLE_K_NYSTROM = 1,       /* affinity-fused Nystrom extension GEMM (Phi)              */
    NLE_K_SINKHORN_PASS = 2, /* one Sinkhorn half-iteration over all local pixels        */
    NLE_K_REDUCE = 3,        /* second-stage fp64 reduction of block partials            */
    NLE_K_GRAM = 4,          /* Gram contraction                                         */
    NLE_K_PROJECT = 5,       /* projection GEMM  V = diag(c) Phi C                       */
    NLE_K_APPLY_REDUCE = 6,  /* t = V^T x                                                */
    NLE_K_APPLY_EXPAND = 7,  /* y_l = V (g_l o t)                                        */
    NLE_K_SMALL = 8,         /* everything p/r/K-sized on the device                     */
    NLE_K_SINK_TABLES = 9,   /* table pass: per-row g tables (k_hist_g)                  */
    NLE_K_GRAM_ROWS = 10,    /* table Gram: per-row histograms (k_ghist_rows)            */
    NLE_K_GRAM_GEMM = 11,    /* table Gram: fp64 MFMA GEMM (k_ghist_gemm)                */
    NLE_KERNEL_COUNT = 12
};
const char* nle_kernel_name(int kid);
/* enable != 0: bracket kernel launches of the ctx with HIP events recorded on the ctx's stream and
 * accumulate per-kernel launch counts and durations (resolved whenever the pipeline synchronises the
 * stream).  enable == 1: the N-sized kernels only; enable == 2: also the p-sized / second-stage
 * kernels (every timed launch opens a ~10 us gap on the stream).  Resets the counters. */
int nle_ctx_profile(nle_ctx* ctx, int enable);
int nle_ctx_kernel_stats(nle_ctx* ctx, int kid, long long* launches, double* total_ms);
/* Run `reps` launches of the materialising affinity kernel (the HBM-roofline pass of
 * computeKernel) and return the average launch duration in ms, measured with HIP events on
 * the ctx's stream. */
int nle_bench_affinity(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples,
                       int n_col_samples, double hx, double hy, float* d_kab, int reps,
                       double* h_avg_ms);
/* the same for the fp64 affinity rows of NLE_MODE_STREAMED_F64 / nle_compute_kernel64 (libm exp, 8 bytes per entry) on the
 * first `rows` image rows of this rank's slab: d_kab holds rows * W * nle_ld(p) doubles */
int nle_bench_affinity64(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                         double hy, long long rows, double* d_kab, int reps, double* h_avg_ms);
/* the range of 16-level tiles [first, first + n) that occur in the plane the filter was trained on: the columns of the
 * look-up-table formulation's g / h tables that its kernels make, store and contract (the others are skipped); (0, 16)
 * for a filter that does not run on level-sorted rows.  For byte models of those kernels. */
int nle_filter_level_tiles(const nle_filter* f, int* first_tile, int* n_tiles);
/* same for one Sinkhorn half-iteration pass over a device-resident phi */
int nle_bench_sinkhorn_pass(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r,
                            int reps, double* h_avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* NLE_H */
