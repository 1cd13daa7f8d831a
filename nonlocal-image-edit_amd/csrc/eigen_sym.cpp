// Host fp64 symmetric eigensolver: Householder tridiagonalisation with accumulated
// transforms, then implicit-shift QL (the classic EISPACK tred2/tql2 scheme), written
// for column-major storage so every inner loop is unit stride.
//
// Replaces Eigen::SelfAdjointEigenSolver at the reference's call site
// src/filter.cpp:207-210 (Eigen3 is a system dependency of the reference, version floor
// 3.3, not vendored).  Same contract: symmetric input, LOWER triangle referenced,
// orthonormal eigenvectors; `eigen_decomposition` adds the reference's post-processing
// (:209-216): descending order, keep the leading run with D >= eps.
#include "eigen_sym.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

namespace nleh {

namespace {

inline double& at(double* v, int n, int row, int col) { return v[(size_t)col * n + row]; }

// hypot without the libm call on the common path (no overflow/underflow of the squares)
inline double hyp(double a, double b) {
    const double s = a * a + b * b;
    return (s > 1e-280 && s < 1e280) ? std::sqrt(s) : std::hypot(a, b);
}

// V: n x n col-major, in: symmetric matrix (lower triangle valid, mirrored by caller);
// out: orthogonal Q with Q^T A Q tridiagonal (d diag, e[1..n-1] sub-diagonal).
// The O(n^3) loops below are unit-stride; the library is built without -march (it has to run on
// whatever host the GPU box has), so they are multiversioned and resolved at load time.
#define NLE_SIMD_CLONES __attribute__((target_clones("default", "avx2", "avx512f")))

NLE_SIMD_CLONES void tridiagonalize(int n, double* V, double* d, double* e) {
    for (int j = 0; j < n; ++j) d[j] = at(V, n, n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) {
                d[j] = at(V, n, i - 1, j);
                at(V, n, i, j) = 0.0;
                at(V, n, j, i) = 0.0;
            }
        } else {
            for (int k = 0; k < i; ++k) {
                d[k] /= scale;
                h += d[k] * d[k];
            }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                at(V, n, j, i) = f;
                g = e[j] + at(V, n, j, j) * f;
                const double* col = &at(V, n, 0, j);
                double gs = 0.0;
#pragma omp simd reduction(+ : gs)
                for (int k = j + 1; k <= i - 1; ++k) {
                    gs += col[k] * d[k];
                    e[k] += col[k] * f;
                }
                e[j] = g + gs;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) {
                e[j] /= h;
                f += e[j] * d[j];
            }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j];
                g = e[j];
                double* col = &at(V, n, 0, j);
#pragma omp simd
                for (int k = j; k <= i - 1; ++k) col[k] -= (f * e[k] + g * d[k]);
                d[j] = at(V, n, i - 1, j);
                at(V, n, i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    // accumulate transformations
    for (int i = 0; i < n - 1; ++i) {
        at(V, n, n - 1, i) = at(V, n, i, i);
        at(V, n, i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            const double* ci1 = &at(V, n, 0, i + 1);
            for (int k = 0; k <= i; ++k) d[k] = ci1[k] / h;
            for (int j = 0; j <= i; ++j) {
                double* cj = &at(V, n, 0, j);
                double g = 0.0;
#pragma omp simd reduction(+ : g)
                for (int k = 0; k <= i; ++k) g += ci1[k] * cj[k];
#pragma omp simd
                for (int k = 0; k <= i; ++k) cj[k] -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) at(V, n, k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) {
        d[j] = at(V, n, n - 1, j);
        at(V, n, n - 1, j) = 0.0;
    }
    at(V, n, n - 1, n - 1) = 1.0;
    e[0] = 0.0;
}

// implicit QL on (d, e) accumulating rotations into V's columns; returns false if an
// eigenvalue needs more than 60 sweeps.
NLE_SIMD_CLONES bool ql_implicit(int n, double* V, double* d, double* e) {
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::ldexp(1.0, -52);
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) {
            if (std::fabs(e[m]) <= eps * tst1) break;
            ++m;
        }
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 60) return false;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hyp(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                const double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hyp(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    double* vi = &at(V, n, 0, i);
                    double* vi1 = &at(V, n, 0, i + 1);
#pragma omp simd
                    for (int k = 0; k < n; ++k) {
                        const double hk = vi1[k];
                        vi1[k] = s * vi[k] + c * hk;
                        vi[k] = c * vi[k] - s * hk;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    return true;
}

// right-looking Cholesky on the lower triangle, column-major (all inner loops unit stride)
NLE_SIMD_CLONES bool cholesky_lower(int n, double* A) {
    for (int j = 0; j < n; ++j) {
        const double djj = at(A, n, j, j);
        if (!(djj > 0.0)) return false;
        const double d = std::sqrt(djj), inv = 1.0 / d;
        double* cj = &at(A, n, 0, j);
        cj[j] = d;
#pragma omp simd
        for (int i = j + 1; i < n; ++i) cj[i] *= inv;
        for (int k = j + 1; k < n; ++k) {
            const double l = cj[k];
            double* ck = &at(A, n, 0, k);
#pragma omp simd
            for (int i = k; i < n; ++i) ck[i] -= l * cj[i];
        }
    }
    return true;
}

// X = L^-1 by column-oriented forward substitution (X lower triangular)
NLE_SIMD_CLONES void lower_inverse(int n, const double* L, double* X) {
    for (int j = 0; j < n; ++j) {
        double* x = X + (size_t)j * n;
        for (int i = 0; i < n; ++i) x[i] = 0.0;
        x[j] = 1.0;
        for (int k = j; k < n; ++k) {
            const double* lk = L + (size_t)k * n;
            const double xk = x[k] / lk[k];
            x[k] = xk;
#pragma omp simd
            for (int i = k + 1; i < n; ++i) x[i] -= xk * lk[i];
        }
    }
}

}  // namespace

NLE_SIMD_CLONES void gemm_nn_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1) {
    (void)n;
    for (int j = j0; j < j1; ++j) {
        double* cc = C + (size_t)j * m;
        std::fill(cc, cc + m, 0.0);
        int l = 0;
        for (; l + 3 < k; l += 4) {  // four columns of A per sweep over the output column
            const double b0 = B[(size_t)j * k + l], b1 = B[(size_t)j * k + l + 1], b2 = B[(size_t)j * k + l + 2],
                         b3 = B[(size_t)j * k + l + 3];
            const double *a0 = A + (size_t)l * m, *a1 = a0 + m, *a2 = a1 + m, *a3 = a2 + m;
#pragma omp simd
            for (int i = 0; i < m; ++i) cc[i] += (a0[i] * b0 + a1[i] * b1) + (a2[i] * b2 + a3[i] * b3);
        }
        for (; l < k; ++l) {
            const double b = B[(size_t)j * k + l];
            const double* a = A + (size_t)l * m;
#pragma omp simd
            for (int i = 0; i < m; ++i) cc[i] += a[i] * b;
        }
    }
}

NLE_SIMD_CLONES void gemm_nt_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
        double* cc = C + (size_t)j * m;
        std::fill(cc, cc + m, 0.0);
        int l = 0;
        for (; l + 3 < k; l += 4) {
            const double b0 = B[(size_t)l * n + j], b1 = B[(size_t)(l + 1) * n + j], b2 = B[(size_t)(l + 2) * n + j],
                         b3 = B[(size_t)(l + 3) * n + j];
            const double *a0 = A + (size_t)l * m, *a1 = a0 + m, *a2 = a1 + m, *a3 = a2 + m;
#pragma omp simd
            for (int i = 0; i < m; ++i) cc[i] += (a0[i] * b0 + a1[i] * b1) + (a2[i] * b2 + a3[i] * b3);
        }
        for (; l < k; ++l) {
            const double b = B[(size_t)l * n + j];
            const double* a = A + (size_t)l * m;
#pragma omp simd
            for (int i = 0; i < m; ++i) cc[i] += a[i] * b;
        }
    }
}

NLE_SIMD_CLONES void gemm_tn_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1) {
    (void)n;
    for (int j = j0; j < j1; ++j)
        for (int l = 0; l < k; ++l) {
            const double* a = A + (size_t)l * m;
            const double* b = B + (size_t)j * m;
            double s = 0.0;
#pragma omp simd reduction(+ : s)
            for (int i = 0; i < m; ++i) s += a[i] * b[i];
            C[(size_t)j * k + l] = s;
        }
}

bool cholesky_with_inverse(const double* M, int n, double* L, double* Linv, double* inv_trace) {
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) L[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : 0.0;
    if (!cholesky_lower(n, L)) return false;
    lower_inverse(n, L, Linv);
    double t = 0.0;
    for (size_t i = 0; i < (size_t)n * n; ++i) t += Linv[i] * Linv[i];
    *inv_trace = t;
    return std::isfinite(t);
}

bool sym_eigen(const double* M, int n, double* U, double* D) {
    if (n <= 0) return true;
    // mirror the lower triangle (SelfAdjointEigenSolver reads only the lower one)
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) {
            const double v = (r >= c) ? M[(size_t)c * n + r] : M[(size_t)r * n + c];
            U[(size_t)c * n + r] = v;
        }
    std::vector<double> e(n);
    if (n == 1) {
        D[0] = U[0];
        U[0] = 1.0;
        return true;
    }
    tridiagonalize(n, U, D, e.data());
    if (!ql_implicit(n, U, D, e.data())) return false;
    // ascending order
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return D[a] < D[b]; });
    std::vector<double> Ds(n), Us((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        Ds[j] = D[idx[j]];
        std::copy(U + (size_t)idx[j] * n, U + (size_t)idx[j] * n + n, Us.begin() + (size_t)j * n);
    }
    std::copy(Ds.begin(), Ds.end(), D);
    std::copy(Us.begin(), Us.end(), U);
    return true;
}

bool eigen_decomposition(const double* M, int n, double eps, double* U, double* D, int* r_out) {
    std::vector<double> Ua((size_t)n * n), Da(n);
    if (!sym_eigen(M, n, Ua.data(), Da.data())) return false;
    // reference src/filter.cpp:209-216: reverse to descending, keep leading run >= eps
    int r = 0;
    for (int j = 0; j < n; ++j) {
        const int src = n - 1 - j;
        D[j] = Da[src];
        std::copy(Ua.begin() + (size_t)src * n, Ua.begin() + (size_t)src * n + n, U + (size_t)j * n);
    }
    while (r < n && D[r] >= eps) ++r;
    *r_out = r;
    return true;
}

}  // namespace nleh
