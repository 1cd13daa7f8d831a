// Host fp64 symmetric eigensolver (see eigen_sym.cpp).  Internal to libnle_hip.so.
#pragma once

#include <functional>

namespace nleh {

// M: n x n column-major symmetric (lower triangle read).  U: n x n column-major
// eigenvectors (columns), D: eigenvalues ASCENDING.  false = QL did not converge.
bool sym_eigen(const double* M, int n, double* U, double* D);

// Same results as sym_eigen (ASCENDING); the QL rotations are recorded and applied to the accumulated orthogonal
// factor by cache-resident row blocks (n = 200: 1.65 -> ~1.45 ms).
bool sym_eigen_blocked(const double* M, int n, double* U, double* D);

// The reference's eigenDecomposition (src/filter.cpp:204-228): eigenvalues DESCENDING,
// *r = length of the leading run with D >= eps (columns/values beyond *r are still
// filled but not part of the result).
bool eigen_decomposition(const double* M, int n, double eps, double* U, double* D, int* r);

// Same, but only the eigenvectors of the `kmax` largest eigenvalues are formed (U: n x min(kmax, n));
// D still receives all n eigenvalues.  What orthogonalize needs (:314 keeps nEigVectors columns).
bool eigen_decomposition_top(const double* M, int n, double eps, int kmax, double* U, double* D, int* r);

// All eigenvalues DESCENDING in D and the first `ncols` eigenvectors in U (n x ncols); nthreads <= 0 picks the
// default (1; NLE_EIG_THREADS overrides).  The rotation and back-transformation phases need no barriers and
// can run on short-lived threads.
bool sym_eigen_top(const double* M, int n, int ncols, int nthreads, double* U, double* D);
// the same for a matrix already reduced to tridiagonal form (V, hs: the Householder vectors and scales; d, e: T), e.g.
// by the device kernel of tridiag.hip; and eigen_decomposition_top on top of it
bool sym_eigen_top_reduced(int n, const double* V, const double* d, const double* e, const double* hs, int ncols,
                           int nthreads, double* U, double* D);
bool eigen_decomposition_top_reduced(int n, double eps, int kmax, const double* V, const double* d, const double* e,
                                     const double* hs, double* U, double* D, int* r);

// All n eigenvalues DESCENDING in D and the eigenvectors of D[first .. first + count) in U (n x count), the latter by
// inverse iteration on the tridiagonal form (cheap when count << n).  false: no convergence.
// kept_out != null: the selection is instead "every eigenvalue after the leading run >= below_eps", made only if there
// are at most max_below of them (*kept_out = length of that run; U must hold n x max_below).
bool sym_eigen_select(const double* M, int n, double* D, int first, int count, double* U, double below_eps = 0.0,
                      int max_below = 0, int* kept_out = nullptr);

// The same by bisection (Sturm counts on the tridiagonal form) for callers that use only the count of the cut, the few
// eigenvalues below it, the largest and the smallest kept one: *kept_out = number of eigenvalues >= eps; if 1 .. max_below
// lie below, Dbelow (DESCENDING) and U (n x max_below) hold them and their eigenvectors.
bool sym_eigen_below(const double* M, int n, double eps, int max_below, int* kept_out, double* lam_max, double* lam_min_kept,
                     double* Dbelow, double* U);
// eigen_decomposition_top computing only what orthogonalize uses of Q (src/filter.cpp:313-316): Dk[0 .. kmax) the kmax
// largest eigenvalues, U (n x kmax) their eigenvectors, *r = number of eigenvalues >= eps -- eigenvalues by bisection
// when 2 kmax <= n (otherwise, or with NLE_EIG_NO_BISECT set, through eigen_decomposition_top).
bool eigen_decomposition_topk(const double* M, int n, double eps, int kmax, double* U, double* Dk, int* r);

// Eigenvectors of the symmetric tridiagonal T (d[0..n) diagonal, e[1..n) sub-diagonal) for its eigenvalues
// lam_all[first .. first + count) (lam_all: all n eigenvalues DESCENDING, accurate to rounding): inverse iteration
// (dstein's scheme), falling back to the QL iteration with accumulated rotations if a vector does not converge.
// Z: n x count column-major.  What the device reduction (dense64.hip: sytrd_dist + tridiag_bisect) leaves to the host.
bool tridiag_eigenvectors(int n, const double* d, const double* e, const double* lam_all, int first, int count, double* Z);

// Opt-in top-K solver with the semantics of the reference's USE_SPECTRA build (src/filter.cpp:170-199; see eigen_sym.cpp):
// the nev = min(nev_in, n - 1) eigenpairs of LARGEST MAGNITUDE of the FULL n x n matrix A (column-major), Krylov
// dimension min(2 nev, n), residual tolerance `tol`, at most `max_restarts` restarts.  Returns the number of converged
// pairs (-1: the projected eigenproblem failed), U: n x nev column-major, D sorted by algebraic value DESCENDING.
int lanczos_topk(const double* A, int n, int nev_in, double tol, int max_restarts, double* U, double* D, int* restarts_out);

// Cholesky factor of a symmetric positive definite matrix (lower triangle of M read): L is n x n
// column-major lower triangular (zeros above the diagonal), Linv = L^-1 likewise.  Returns false if
// a pivot is not positive.  *inv_trace = trace(M^-1) = ||Linv||_F^2, so every eigenvalue of M is at
// least 1 / *inv_trace: the caller's proof that the reference's eigenvalue cut at eps removes nothing.
// max_inv_trace > 0: the caller will reject the factor if trace(M^-1) exceeds it -- the factorisation then stops at the
// first pivot that already proves that (returns false)
bool cholesky_with_inverse(const double* M, int n, double* L, double* Linv, double* inv_trace, double max_inv_trace = 0.0);

// Column ranges [j0, j1) of small column-major products (the caller splits columns over threads):
//   nn: C (m x n) = A (m x k) B (k x n);  nt: C (m x n) = A (m x k) B^T (B n x k);  tn: C (k x n) = A^T B (A m x k, B m x n)
void gemm_nn_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1);
void gemm_nt_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1);
void gemm_tn_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1);

// body(part) for part in [0, nparts) on up to nthreads threads (the caller's plus short-lived helpers pinned to its
// L3 domain); parts are dealt round-robin
void run_parts(int nparts, int nthreads, const std::function<void(int)>& body);

}  // namespace nleh
