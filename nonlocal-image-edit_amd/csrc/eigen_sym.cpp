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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

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

// ---- the same solver in three barrier-free phases (what sym_eigen_top runs)
//   1. tridiag_reduce: the reduction half of `tridiagonalize` (Householder vectors u_i stay in column i,
//      rows 0..i-1, with scale h_i in hs[i]); the orthogonal factor is never formed.
//   2. ql_record: implicit QL on (d, e) alone, RECORDING the plane rotations of every sweep.
//   3. The rotations are applied to Z = I by independent row blocks (one pass over a block per sweep, the
//      shared column of consecutive rotations carried in registers), then only the wanted columns of Z are
//      back-transformed with the Householder vectors, by independent column blocks.
// Phases 3a/3b split over threads without any synchronisation, which the per-sweep update of the classic
// loop does not allow.
// (v8d: eight doubles; defined here because the two inner loops below are written on it)
typedef double v8d __attribute__((vector_size(64)));
static inline __attribute__((always_inline)) v8d ld8(const double* p) {
    v8d v;
    __builtin_memcpy(&v, p, sizeof(v));
    return v;
}
static inline __attribute__((always_inline)) void st8(double* p, v8d v) { __builtin_memcpy(p, &v, sizeof(v)); }
static inline __attribute__((always_inline)) double hsum8(v8d v) {
    return ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
}

// Householder reduction on the UPPER triangle, column-major, eight columns per sweep.  Step i annihilates column i above the
// sub-diagonal entry: the vector x = A(0..i-1, i), the columns' active parts A(0..j, j) and the stored reflector u_i (column
// i, rows < i) are all CONTIGUOUS -- the row-oriented classic (tred2) gathers and scatters a strided row every step.  The two
// O(i^2) loops of a step, p = A u and A -= u q^T + q u^T, take eight columns at a time over full vectors of eight rows; where
// a block of columns crosses the diagonal the 8 x 8 triangle is one more vector per column under a 0/1 mask, and the
// work vectors are zero beyond i: no scalar remainder loops anywhere (they, the strided copies and the scalar O(i) loops
// were two thirds of the four-column form's 0.33 ms at n = 200).  Reads the whole of V (both triangles mirrored by the
// caller); on return column i, rows < i, holds u_i with H_i = I - u_i u_i^T / hs[i], d and e the tridiagonal matrix
// (e[i] couples i - 1 and i): the layout back_transform_cols and the tridiagonal routines take.
NLE_SIMD_CLONES static void tridiag_reduce_impl(int n, double* V, double* d, double* e, double* hs, const double (*kLe)[8], const double (*kLt)[8]) {
    // The rank-2 update of step i+1 and the product A u of step i are ONE sweep over the block: an entry is loaded, updated,
    // stored and multiplied while it is in a register (the two-sweep form loaded it twice; same operations on the same operands
    // in the same order, bit for bit the same T).  Column i itself is brought up to date first (O(i)): its reflector must be
    // known before the sweep starts.
    std::vector<double> work((size_t)4 * (n + 16), 0.0);
    double* u = work.data();       // the pending update's vectors (step i+1), zero from i+1 on; zero at the start
    double* q = u + (n + 16);
    double* un = q + (n + 16);     // this step's reflector and product
    double* qn = un + (n + 16);
    for (int i = n - 1; i >= 0; --i) {
        double* x = V + (size_t)i * n;
        const int i8 = (i + 7) & ~7;
        if (i < n - 1) {  // column i and its diagonal entry after the pending update (u, q are zero beyond row i: the rows below
                          // the diagonal, and the entries of the next column a last vector reaches, are rewritten unchanged)
            const double ui = u[i], qi = q[i];
            for (int k = 0; k <= i; k += 8) st8(x + k, ld8(x + k) - (ld8(u + k) * qi + ld8(q + k) * ui));
        }
        if (i == 0) break;
        double scale = 0.0, h = 0.0;
#pragma omp simd reduction(+ : scale)
        for (int k = 0; k < i; ++k) scale += std::fabs(x[k]);
        for (int k = 0; k < i8 + 8; ++k) un[k] = 0.0;
        for (int k = 0; k < i8 + 8; ++k) qn[k] = 0.0;
        if (scale == 0.0) {
            e[i] = x[i - 1];
            for (int k = 0; k < i; ++k) x[k] = 0.0;
        } else {
            const double rscale = 1.0 / scale;
#pragma omp simd reduction(+ : h)
            for (int k = 0; k < i; ++k) {
                const double t = x[k] * rscale;
                un[k] = t;
                h += t * t;
            }
            const double f0 = un[i - 1];
            double g = std::sqrt(h);
            if (f0 > 0) g = -g;
            e[i] = scale * g;
            h -= f0 * g;
            un[i - 1] = f0 - g;
#pragma omp simd
            for (int k = 0; k < i; ++k) x[k] = un[k];
        }
        // the block's upper triangle: A -= u q^T + q u^T (pending), then qn = A un on what was just stored
        for (int j0 = 0; j0 < i; j0 += 8) {
            const int nc = std::min(8, i - j0);
            double* col[8];
            double uj[8], qj[8], unj[8];
            for (int c = 0; c < 8; ++c) {
                col[c] = V + (size_t)(j0 + (c < nc ? c : 0)) * n;
                uj[c] = c < nc ? u[j0 + c] : 0.0;
                qj[c] = c < nc ? q[j0 + c] : 0.0;
                unj[c] = c < nc ? un[j0 + c] : 0.0;
            }
            v8d acc[8];
            for (int c = 0; c < 8; ++c) acc[c] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < j0; k += 8) {
                const v8d uv = ld8(u + k), qv = ld8(q + k), nv = ld8(un + k);
                v8d pv = ld8(qn + k);
                for (int c = 0; c < 8; ++c) {
                    const v8d cv = ld8(col[c] + k) - (uv * qj[c] + qv * uj[c]);
                    if (c < nc) st8(col[c] + k, cv);
                    acc[c] += cv * nv;
                    pv += cv * unj[c];
                }
                st8(qn + k, pv);
            }
            {
                const v8d uv = ld8(u + j0), qv = ld8(q + j0), nv = ld8(un + j0);
                v8d pv = ld8(qn + j0);
                for (int c = 0; c < 8; ++c) {
                    const v8d cv = ld8(col[c] + j0) - ld8(kLe[c]) * (uv * qj[c] + qv * uj[c]);
                    if (c < nc) st8(col[c] + j0, cv);
                    acc[c] += (cv * ld8(kLe[c])) * nv;
                    pv += (cv * ld8(kLt[c])) * unj[c];
                }
                st8(qn + j0, pv);
            }
            for (int c = 0; c < nc; ++c) qn[j0 + c] += hsum8(acc[c]);
        }
        if (scale != 0.0) {
            const double rh = 1.0 / h;
            double f = 0.0;
#pragma omp simd reduction(+ : f)
            for (int k = 0; k < i; ++k) {
                qn[k] *= rh;
                f += qn[k] * un[k];
            }
            const double hh = f / (h + h);
#pragma omp simd
            for (int k = 0; k < i; ++k) qn[k] -= hh * un[k];
            for (int k = i; k < i8 + 8; ++k) qn[k] = 0.0;
        } else {
            for (int k = 0; k < i8 + 8; ++k) qn[k] = 0.0;  // no reflector: nothing pending for the next step
        }
        hs[i] = h;
        std::swap(u, un);
        std::swap(q, qn);
    }
    hs[0] = 0.0;
    for (int j = 0; j < n; ++j) d[j] = at(V, n, j, j);  // the diagonal of T
    e[0] = 0.0;
}

void tridiag_reduce(int n, double* V, double* d, double* e, double* hs) {
    static const double kLe[8][8] = {{1, 0, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 0, 0, 0, 0},
                                     {1, 1, 1, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 1, 1, 0, 0}, {1, 1, 1, 1, 1, 1, 1, 0}, {1, 1, 1, 1, 1, 1, 1, 1}};
    static const double kLt[8][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {1, 0, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 0, 0, 0, 0, 0},
                                     {1, 1, 1, 1, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 1, 1, 0, 0}, {1, 1, 1, 1, 1, 1, 1, 0}};
    if (n < 16) {  // the sweeps read (and rewrite unchanged) up to seven entries past a column's end: give a tiny matrix room
        std::vector<double> Vp((size_t)n * n + 16, 0.0);
        std::copy(V, V + (size_t)n * n, Vp.begin());
        tridiag_reduce_impl(n, Vp.data(), d, e, hs, kLe, kLt);
        std::copy(Vp.begin(), Vp.begin() + (size_t)n * n, V);
        return;
    }
    tridiag_reduce_impl(n, V, d, e, hs, kLe, kLt);
}

struct Sweep {
    int l, m;      // rotations act on columns (i, i+1) for i = m-1 .. l
    size_t first;  // index of the i = m-1 rotation in the (c, s) lists
};

// RECORD == false: the same iteration (bit-identical eigenvalues) without keeping the rotations -- for callers that take
// their eigenvectors from inverse iteration
template <bool RECORD>
bool ql_iterate(int n, double* d, double* e, std::vector<Sweep>& sweeps, std::vector<double>& cs, std::vector<double>& sn) {
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
                if (RECORD) sweeps.push_back(Sweep{l, m, cs.size()});
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
                    if (RECORD) {
                        cs.push_back(c);
                        sn.push_back(s);
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

bool ql_record(int n, double* d, double* e, std::vector<Sweep>& sweeps, std::vector<double>& cs, std::vector<double>& sn) {
    return ql_iterate<true>(n, d, e, sweeps, cs, sn);
}
// Eigenvalues only, by the square-root-free QL variant of Pal, Walker and Kahan as LAPACK's dsterf organises it (implicit
// Wilkinson shift; deflation where e_m^2 <= eps^2 |d_m d_{m+1}|; 2 x 2 blocks in closed form, dlae2): one reciprocal
// and a handful of multiplications per rotation instead of tql2's two hypot calls -- 0.22 -> ~0.1 ms at n = 196.
// d: diagonal, e[1..n): sub-diagonal (e[i] couples i-1 and i) on entry; d holds the eigenvalues (unsorted) on return.
bool ql_values(int n, double* d, double* e2) {
    if (n <= 1) return true;
    if (std::getenv("NLE_EIG_TQL") != nullptr) {  // the rotation-based iteration (what ql_record runs)
        std::vector<Sweep> sweeps;
        std::vector<double> cs, sn;
        return ql_iterate<false>(n, d, e2, sweeps, cs, sn);
    }
    const double eps = std::ldexp(1.0, -53), eps2 = eps * eps, safmin = 2.2250738585072014e-308;
    double anorm = 0.0;
    for (int i = 0; i < n; ++i) anorm = std::max(anorm, std::fabs(d[i]) + (i + 1 < n ? std::fabs(e2[i + 1]) : 0.0) + std::fabs(e2[i]));
    if (anorm == 0.0) return true;
    // scale to norm 1 (squares of tiny off-diagonals must not underflow), e2[i] := (e[i+1] / anorm)^2 couples i and i+1
    const double scl = 1.0 / anorm;
    for (int i = 0; i < n; ++i) d[i] *= scl;
    for (int i = 0; i + 1 < n; ++i) {
        const double t = e2[i + 1] * scl;
        e2[i] = t * t;
    }
    e2[n - 1] = 0.0;
    int iter_left = 60 * n;
    for (int l = 0; l < n; ++l) {
        for (;;) {
            int m = l;
            while (m + 1 < n) {
                if (e2[m] <= eps2 * std::fabs(d[m] * d[m + 1]) + safmin) break;
                ++m;
            }
            if (m + 1 < n) e2[m] = 0.0;  // the block [l, m] is decoupled from the rest
            if (m == l) break;           // d[l] is an eigenvalue
            if (m == l + 1) {   // 2 x 2 block [[a, b], [b, c]]: dlae2
                const double a = d[l], c = d[l + 1], b = std::sqrt(e2[l]);
                const double sm = a + c, df = a - c, adf = std::fabs(df), tb = b + b, ab = std::fabs(tb);
                const double acmx = std::fabs(a) > std::fabs(c) ? a : c, acmn = std::fabs(a) > std::fabs(c) ? c : a;
                double rt;
                if (adf > ab) rt = adf * std::sqrt(1.0 + (ab / adf) * (ab / adf));
                else if (adf < ab) rt = ab * std::sqrt(1.0 + (adf / ab) * (adf / ab));
                else rt = ab * std::sqrt(2.0);
                double rt1, rt2;
                if (sm < 0.0) {
                    rt1 = 0.5 * (sm - rt);
                    rt2 = (acmx / rt1) * acmn - (b / rt1) * b;
                } else if (sm > 0.0) {
                    rt1 = 0.5 * (sm + rt);
                    rt2 = (acmx / rt1) * acmn - (b / rt1) * b;
                } else {
                    rt1 = 0.5 * rt;
                    rt2 = -0.5 * rt;
                }
                d[l] = rt1;
                d[l + 1] = rt2;
                e2[l] = 0.0;
                ++l;  // both are done
                break;
            }
            if (--iter_left < 0) return false;
            // Wilkinson shift from the leading 2 x 2 block of [l, m]
            const double rte = std::sqrt(e2[l]);
            const double p0 = d[l];
            double sigma = (d[l + 1] - p0) / (2.0 * rte);
            const double r0 = hyp(sigma, 1.0);
            sigma = p0 - rte / (sigma + (sigma >= 0.0 ? r0 : -r0));
            double c = 1.0, sn = 0.0, gamma = d[m] - sigma, p = gamma * gamma;
            for (int i = m - 1; i >= l; --i) {
                const double bb = e2[i], r = p + bb;
                if (i != m - 1) e2[i + 1] = sn * r;
                const double oldc = c, rinv = 1.0 / r;
                c = p * rinv;
                sn = bb * rinv;
                const double oldgam = gamma, alpha = d[i];
                gamma = c * (alpha - sigma) - sn * oldgam;
                d[i + 1] = oldgam + (alpha - gamma);
                p = (c != 0.0) ? (gamma * gamma) / c : oldc * bb;
            }
            e2[l] = sn * p;
            d[l] = sigma + gamma;
        }
    }
    for (int i = 0; i < n; ++i) d[i] *= anorm;
    return true;
}

// Rows [k0, k0 + 8 NV) of Z (column-major, leading dimension ldz, a multiple of 8): every sweep's rotations,
// in order.  Consecutive rotations share a column; its running value stays in NV vector registers, so each
// column is loaded and stored once per sweep, and NV independent dependency chains hide the FMA latency.

template <int NV>
static inline __attribute__((always_inline)) void rotate_rows(int ldz, double* Z, int k0, const Sweep* sweeps,
                                                              size_t nsweeps, const double* cs, const double* sn) {
    for (size_t q = 0; q < nsweeps; ++q) {
        const Sweep sw = sweeps[q];
        const double* c = cs + sw.first;
        const double* s = sn + sw.first;
        v8d carry[NV];
        const double* top = Z + (size_t)sw.m * ldz + k0;
#pragma GCC unroll 8
        for (int v = 0; v < NV; ++v) carry[v] = ld8(top + 8 * v);
        for (int i = sw.m - 1, t = 0; i >= sw.l; --i, ++t) {
            const double ci = c[t], si = s[t];
            double* vi = Z + (size_t)i * ldz + k0;
            double* vi1 = vi + ldz;
#pragma GCC unroll 8
            for (int v = 0; v < NV; ++v) {
                const v8d hk = carry[v], vk = ld8(vi + 8 * v);
                st8(vi1 + 8 * v, si * vk + ci * hk);
                carry[v] = ci * vk - si * hk;
            }
        }
        double* bot = Z + (size_t)sw.l * ldz + k0;
#pragma GCC unroll 8
        for (int v = 0; v < NV; ++v) st8(bot + 8 * v, carry[v]);
    }
}

// Z starts as the identity, so entries far from the diagonal are products of many sines and pass through the
// denormal range on their way to zero; denormal operands cost ~100 cycles each on x86.  They are flushed
// to zero for the duration of the phase (a change below 1e-307 in an orthogonal matrix).
struct FlushDenormals {
#if defined(__x86_64__)
    unsigned saved = _mm_getcsr();
    FlushDenormals() { _mm_setcsr(saved | 0x8040); }  // FTZ | DAZ
    ~FlushDenormals() { _mm_setcsr(saved); }
#endif
};

NLE_SIMD_CLONES void apply_rotations_rows(int ldz, double* Z, int k0, int nvec, const Sweep* sweeps, size_t nsweeps,
                                          const double* cs, const double* sn) {
    FlushDenormals ftz;
    while (nvec > 0) {  // 8, 4, 2, 1 vectors of 8 rows
        if (nvec >= 8) {
            rotate_rows<8>(ldz, Z, k0, sweeps, nsweeps, cs, sn);
            nvec -= 8, k0 += 64;
        } else if (nvec >= 4) {
            rotate_rows<4>(ldz, Z, k0, sweeps, nsweeps, cs, sn);
            nvec -= 4, k0 += 32;
        } else if (nvec >= 2) {
            rotate_rows<2>(ldz, Z, k0, sweeps, nsweeps, cs, sn);
            nvec -= 2, k0 += 16;
        } else {
            rotate_rows<1>(ldz, Z, k0, sweeps, nsweeps, cs, sn);
            nvec -= 1, k0 += 8;
        }
    }
}

// Y (n x ncols, column-major, ld n) <- Q Y with Q = P_{n-1} ... P_1, P_i = I - u_i u_i^T / h_i on rows 0..i-1
// (u_i = column i of V, rows 0..i-1): the product `tridiagonalize` accumulates, applied to a few columns.
// Eight columns share every load of a reflector (their eight dot products are independent chains; vector by vector the
// loop waited on one horizontal sum and one division per reflector and column): 0.20 -> 0.0x ms for 50 vectors at n = 200.
template <int NB>
static inline __attribute__((always_inline)) void back_transform_block(int n, const double* V, const double* hs, double* Y0, int nb) {
    double* y[NB];
    for (int jj = 0; jj < NB; ++jj) y[jj] = Y0 + (size_t)(jj < nb ? jj : nb - 1) * n;  // (columns >= nb: recomputed, not stored)
    for (int i = 1; i < n; ++i) {
        const double h = hs[i];
        if (h == 0.0) continue;
        const double* u = V + (size_t)i * n;
        const int i8 = i & ~7;
        v8d acc[NB];
        for (int jj = 0; jj < NB; ++jj) acc[jj] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < i8; k += 8) {
            const v8d uv = ld8(u + k);
            for (int jj = 0; jj < NB; ++jj) acc[jj] += uv * ld8(y[jj] + k);
        }
        double g[NB];
        for (int jj = 0; jj < NB; ++jj) g[jj] = hsum8(acc[jj]);
        for (int k = i8; k < i; ++k)
            for (int jj = 0; jj < NB; ++jj) g[jj] += u[k] * y[jj][k];
        const double rh = 1.0 / h;
        for (int jj = 0; jj < NB; ++jj) g[jj] *= rh;
        for (int k = 0; k < i8; k += 8) {
            const v8d uv = ld8(u + k);
            for (int jj = 0; jj < NB; ++jj)
                if (jj < nb) st8(y[jj] + k, ld8(y[jj] + k) - g[jj] * uv);
        }
        for (int k = i8; k < i; ++k)
            for (int jj = 0; jj < nb; ++jj) y[jj][k] -= g[jj] * u[k];
    }
}
NLE_SIMD_CLONES void back_transform_cols(int n, const double* V, const double* hs, double* Y, int j0, int j1) {
    constexpr int CB = 8;
    for (int jb = j0; jb < j1; jb += CB) back_transform_block<CB>(n, V, hs, Y + (size_t)jb * n, std::min(CB, j1 - jb));
}

// CPUs that share the calling thread's last-level cache (Linux sysfs), empty if unknown: where helper threads are
// pinned when NLE_PIN_THREADS is set (left to the scheduler on a 256-CPU host they start on other core complexes,
// with cold caches and a remote copy of the matrix).
std::vector<int> llc_siblings() {
    std::vector<int> cpus;
#if defined(__linux__)
    const int cpu = sched_getcpu();
    if (cpu < 0) return cpus;
    char path[128];
    std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE* fh = std::fopen(path, "r");
    if (!fh) return cpus;
    char buf[512] = {0};
    if (std::fgets(buf, sizeof buf, fh)) {
        const char* q = buf;
        while (*q) {  // "a-b,c,d-e"
            char* end = nullptr;
            long a = std::strtol(q, &end, 10);
            if (end == q) break;
            long b = a;
            if (*end == '-') {
                q = end + 1;
                b = std::strtol(q, &end, 10);
            }
            for (long c = a; c <= b && cpus.size() < 256; ++c) cpus.push_back((int)c);
            q = (*end == ',') ? end + 1 : end;
            if (*end != ',') break;
        }
    }
    std::fclose(fh);
#endif
    return cpus;
}

void pin_to(std::thread& t, const std::vector<int>& cpus) {
#if defined(__linux__)
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
#else
    (void)t;
    (void)cpus;
#endif
}

template <typename F>
void run_split(int nparts, int nthreads, F&& body) {  // body(part) for part in [0, nparts), split over threads
    nthreads = std::max(1, std::min(nthreads, nparts));
    if (nthreads == 1) {
        for (int q = 0; q < nparts; ++q) body(q);
        return;
    }
    static const bool pin = std::getenv("NLE_PIN_THREADS") != nullptr;  // opt-in, see default_threads
    const std::vector<int> near = pin ? llc_siblings() : std::vector<int>();
    std::vector<std::thread> th;
    auto work = [&](int t) {
        for (int q = t; q < nparts; q += nthreads) body(q);
    };
    for (int t = 1; t < nthreads; ++t) {
        th.emplace_back(work, t);
        pin_to(th.back(), near);
    }
    work(0);
    for (auto& x : th) x.join();
}

// right-looking Cholesky on the lower triangle, column-major (all inner loops unit stride)
// min_pivot: give up as soon as a pivot L_jj^2 falls below it (L_jj^2 >= lambda_min of the leading block >= lambda_min(A),
// so a pivot below tau proves trace(A^-1) > 1 / tau: callers that would reject the factor for that anyway stop here
// instead of finishing the factorisation and inverting it)
NLE_SIMD_CLONES bool cholesky_lower(int n, double* A, double min_pivot) {
    for (int j = 0; j < n; ++j) {
        const double djj = at(A, n, j, j);
        if (!(djj > min_pivot)) return false;
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

// C[:, j0:j1) = A * op(B): register-blocked, 16 rows (two 8-wide vectors) x 6 columns of C per micro-tile,
// the k loop innermost (2 vector loads of A and 6 scalar loads of B per 12 FMAs); rows beyond a multiple
// of 16 fall back to 8-row and scalar tiles.  BT selects B^T (B stored n x k) for the nt form.
template <bool BT, int NR>
static inline __attribute__((always_inline)) void gemm_tile16(const double* A, const double* B, double* C, int m, int k,
                                                              int n, int i0, int j) {
    v8d acc0[NR], acc1[NR];
#pragma GCC unroll 6
    for (int q = 0; q < NR; ++q) acc0[q] = acc1[q] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < k; ++l) {
        const v8d a0 = ld8(A + (size_t)l * m + i0), a1 = ld8(A + (size_t)l * m + i0 + 8);
#pragma GCC unroll 6
        for (int q = 0; q < NR; ++q) {
            const double b = BT ? B[(size_t)l * n + j + q] : B[(size_t)(j + q) * k + l];
            acc0[q] += a0 * b;
            acc1[q] += a1 * b;
        }
    }
#pragma GCC unroll 6
    for (int q = 0; q < NR; ++q) {
        st8(C + (size_t)(j + q) * m + i0, acc0[q]);
        st8(C + (size_t)(j + q) * m + i0 + 8, acc1[q]);
    }
}
template <bool BT, int NR>
static inline __attribute__((always_inline)) void gemm_tile8(const double* A, const double* B, double* C, int m, int k,
                                                             int n, int i0, int j) {
    v8d acc[NR];
#pragma GCC unroll 6
    for (int q = 0; q < NR; ++q) acc[q] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < k; ++l) {
        const v8d a0 = ld8(A + (size_t)l * m + i0);
#pragma GCC unroll 6
        for (int q = 0; q < NR; ++q) acc[q] += a0 * (BT ? B[(size_t)l * n + j + q] : B[(size_t)(j + q) * k + l]);
    }
#pragma GCC unroll 6
    for (int q = 0; q < NR; ++q) st8(C + (size_t)(j + q) * m + i0, acc[q]);
}
template <bool BT, int NR>
static inline __attribute__((always_inline)) void gemm_panel(const double* A, const double* B, double* C, int m, int k,
                                                             int n, int j) {
    int i0 = 0;
    for (; i0 + 16 <= m; i0 += 16) gemm_tile16<BT, NR>(A, B, C, m, k, n, i0, j);
    if (i0 + 8 <= m) {
        gemm_tile8<BT, NR>(A, B, C, m, k, n, i0, j);
        i0 += 8;
    }
    for (; i0 < m; ++i0)
        for (int q = 0; q < NR; ++q) {
            double sacc = 0.0;
            for (int l = 0; l < k; ++l) sacc += A[(size_t)l * m + i0] * (BT ? B[(size_t)l * n + j + q] : B[(size_t)(j + q) * k + l]);
            C[(size_t)(j + q) * m + i0] = sacc;
        }
}
template <bool BT>
static inline __attribute__((always_inline)) void gemm_cols(const double* A, const double* B, double* C, int m, int k,
                                                            int n, int j0, int j1) {
    int j = j0;
    for (; j + 6 <= j1; j += 6) gemm_panel<BT, 6>(A, B, C, m, k, n, j);
    for (; j + 2 <= j1; j += 2) gemm_panel<BT, 2>(A, B, C, m, k, n, j);
    for (; j < j1; ++j) gemm_panel<BT, 1>(A, B, C, m, k, n, j);
}

NLE_SIMD_CLONES void gemm_nn_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1) {
    gemm_cols<false>(A, B, C, m, k, n, j0, j1);
}

NLE_SIMD_CLONES void gemm_nt_cols(const double* A, const double* B, double* C, int m, int k, int n, int j0, int j1) {
    gemm_cols<true>(A, B, C, m, k, n, j0, j1);
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

// ---- blocked Cholesky factor and inverse of a p x p matrix on one core (what the train path runs on Ka and on Wa's deflated
// matrix below 288 samples: 0.45 + 0.33 ms of a 7 ms cfg4 step with the rank-1 / column-sweep forms above, which stream the
// matrix from L2 once per column).  Register-blocked tiles (16 rows x 6 columns, the k loop innermost) on explicit leading
// dimensions; triangular operands only trim the k range of a tile.
namespace {
// C[i0 .. i0 + 8 RV) x [j .. j + NR) += sign * sum_{l in [l0, l1)} A(i, l) B(l, j + q);  B(l, c) = TB ? B[c + l ldb] : B[l + c ldb]
template <int RV, int NR, bool TB>
static inline __attribute__((always_inline)) void mk_tile(double sign, const double* A, int lda, const double* B, int ldb, double* C,
                                                          int ldc, int i0, int j, int l0, int l1) {
    v8d acc[RV][NR];
#pragma GCC unroll 2
    for (int r = 0; r < RV; ++r)
#pragma GCC unroll 6
        for (int q = 0; q < NR; ++q) acc[r][q] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = l0; l < l1; ++l) {
        v8d a[RV];
#pragma GCC unroll 2
        for (int r = 0; r < RV; ++r) a[r] = ld8(A + (size_t)l * lda + i0 + 8 * r);
#pragma GCC unroll 6
        for (int q = 0; q < NR; ++q) {
            const double b = TB ? B[(size_t)l * ldb + j + q] : B[(size_t)(j + q) * ldb + l];
#pragma GCC unroll 2
            for (int r = 0; r < RV; ++r) acc[r][q] += a[r] * b;
        }
    }
#pragma GCC unroll 2
    for (int r = 0; r < RV; ++r)
#pragma GCC unroll 6
        for (int q = 0; q < NR; ++q) {
            double* c = C + (size_t)(j + q) * ldc + i0 + 8 * r;
            st8(c, ld8(c) + sign * acc[r][q]);
        }
}
// C (m x n, ldc) += sign * A (m x k, lda) * op(B).  lower_c: only tiles that touch the lower triangle of a square C are
// computed (entries above the diagonal inside such a tile are touched too: the caller does not keep anything there);
// a_lower: A is square lower triangular (A(i, l) = 0 for l > i);  b_lower: op(B) is lower triangular (B(l, c) = 0 for l < c).
template <bool TB>
static inline __attribute__((always_inline)) void mk_gemm(double sign, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                                                         int m, int n, int k, bool lower_c, bool a_lower, bool b_lower) {
    for (int j = 0; j < n; j += 6) {
        const int nr = std::min(6, n - j);
        const int l0 = b_lower ? j : 0;
        int i0 = lower_c ? (j / 16) * 16 : 0;
        for (; i0 < m; i0 += 16) {
            const int rows = std::min(16, m - i0);
            const int l1 = a_lower ? std::min(k, i0 + rows) : k;
            if (l1 <= l0) continue;
            if (rows == 16 && nr == 6) {
                mk_tile<2, 6, TB>(sign, A, lda, B, ldb, C, ldc, i0, j, l0, l1);
            } else if (rows >= 8 && nr == 6) {
                mk_tile<1, 6, TB>(sign, A, lda, B, ldb, C, ldc, i0, j, l0, l1);
                for (int i = i0 + 8; i < i0 + rows; ++i)
                    for (int q = 0; q < nr; ++q) {
                        double sacc = 0.0;
                        for (int l = l0; l < l1; ++l) sacc += A[(size_t)l * lda + i] * (TB ? B[(size_t)l * ldb + j + q] : B[(size_t)(j + q) * ldb + l]);
                        C[(size_t)(j + q) * ldc + i] += sign * sacc;
                    }
            } else {
                for (int i = i0; i < i0 + rows; ++i)
                    for (int q = 0; q < nr; ++q) {
                        double sacc = 0.0;
                        for (int l = l0; l < l1; ++l) sacc += A[(size_t)l * lda + i] * (TB ? B[(size_t)l * ldb + j + q] : B[(size_t)(j + q) * ldb + l]);
                        C[(size_t)(j + q) * ldc + i] += sign * sacc;
                    }
            }
        }
    }
}

// right-looking blocked Cholesky on the lower triangle (column-major, leading dimension n): 32 columns at a time by the
// rank-1 form restricted to the block's columns, the trailing matrix by one tile product per block.  The strict upper
// triangle is used as scratch by the tiles on the diagonal and zeroed at the end.  min_pivot as in cholesky_lower.
NLE_SIMD_CLONES bool cholesky_lower_blocked(int n, double* A, double min_pivot) {
    constexpr int NB = 32;
    if (n < 96) return cholesky_lower(n, A, min_pivot);
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = std::min(NB, n - k0), kend = k0 + kb, m = n - kend;
        for (int j = k0; j < kend; ++j) {
            const double djj = at(A, n, j, j);
            if (!(djj > min_pivot)) return false;
            const double d = std::sqrt(djj), inv = 1.0 / d;
            double* cj = &at(A, n, 0, j);
            cj[j] = d;
#pragma omp simd
            for (int i = j + 1; i < n; ++i) cj[i] *= inv;
            for (int k = j + 1; k < kend; ++k) {
                const double l = cj[k];
                double* ck = &at(A, n, 0, k);
#pragma omp simd
                for (int i = k; i < n; ++i) ck[i] -= l * cj[i];
            }
        }
        if (m > 0)  // A22 -= L21 L21^T, lower tiles only
            mk_gemm<true>(-1.0, &at(A, n, kend, k0), n, &at(A, n, kend, k0), n, &at(A, n, kend, kend), n, m, m, kb, true, false, false);
    }
    for (int c = 1; c < n; ++c)
        for (int r = 0; r < c; ++r) at(A, n, r, c) = 0.0;
    return true;
}

// X = L^-1 (both n x n lower triangular, leading dimensions ldl / ldx; X's block must be zero on entry): halves recursively,
// [L11 0; L21 L22]^-1 = [X11 0; -X22 (L21 X11) X22], the two products on the tile kernel with the triangles' zeros skipped
NLE_SIMD_CLONES void lower_inverse_blocked(int n, const double* L, int ldl, double* X, int ldx, double* tmp) {
    if (n <= 32) {
        for (int j = 0; j < n; ++j) {
            double* x = X + (size_t)j * ldx;
            x[j] = 1.0;
            for (int k = j; k < n; ++k) {
                const double* lk = L + (size_t)k * ldl;
                const double xk = x[k] / lk[k];
                x[k] = xk;
#pragma omp simd
                for (int i = k + 1; i < n; ++i) x[i] -= xk * lk[i];
            }
        }
        return;
    }
    const int n1 = ((n / 2) + 15) & ~15, n2 = n - n1;
    lower_inverse_blocked(n1, L, ldl, X, ldx, tmp);
    lower_inverse_blocked(n2, L + (size_t)n1 * ldl + n1, ldl, X + (size_t)n1 * ldx + n1, ldx, tmp);
    // T (n2 x n1) = L21 X11 ;  X21 = -X22 T
    double* T = tmp;
    for (size_t i = 0; i < (size_t)n2 * n1; ++i) T[i] = 0.0;
    mk_gemm<false>(1.0, L + n1, ldl, X, ldx, T, n2, n2, n1, n1, false, false, true);
    mk_gemm<false>(-1.0, X + (size_t)n1 * ldx + n1, ldx, T, n2, X + n1, ldx, n2, n1, n2, false, true, false);
}
}  // namespace

bool cholesky_with_inverse(const double* M, int n, double* L, double* Linv, double* inv_trace, double max_inv_trace) {
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) L[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : 0.0;
    if (!cholesky_lower_blocked(n, L, max_inv_trace > 0.0 ? 1.0 / max_inv_trace : 0.0)) return false;
    if (n < 96) {
        lower_inverse(n, L, Linv);
    } else {
        for (size_t i = 0; i < (size_t)n * n; ++i) Linv[i] = 0.0;
        std::vector<double> tmp((size_t)(n / 2 + 16) * (n / 2 + 16));
        lower_inverse_blocked(n, L, n, Linv, n, tmp.data());
    }
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

// Threads for the two parallel phases.  Measured on the GPU box's host (EPYC 9575F, a 16-CPU quota over 256 logical
// CPUs shared with other tenants), n = 200: helpers left to the scheduler start on other core complexes and make the
// solve slower (1.3 -> 1.8-2.8 ms); pinned to the caller's L3 domain (NLE_PIN_THREADS=1) they gain a little (all
// eigenvectors 1.90 -> 1.35 ms on four threads, top 50 1.27 -> 1.14 ms on two) but stall for milliseconds whenever
// that domain is busy with somebody else's work, which a shared host cannot rule out.  So: one thread below
// n = 512; from there eight unpinned helpers pay (n = 900, all eigenvectors: 124 -> ~70 ms).  NLE_EIG_THREADS overrides.
void run_parts(int nparts, int nthreads, const std::function<void(int)>& body) { run_split(nparts, nthreads, body); }

int default_threads(int n, int) {
    if (const char* e = std::getenv("NLE_EIG_THREADS")) return std::max(1, std::atoi(e));
    return n >= 512 ? 8 : 1;
}

// x <- x minus its components along the m orthonormal vectors z_0 .. z_{m-1} (stride n), two passes.  Four vectors at a
// time: their four dot products in ONE sweep over x (independent accumulators), their combined update in one more -- half the
// sweeps of vector-by-vector modified Gram-Schmidt and vectorisable; inside a block the projection is classical, across
// blocks sequential, and the second pass makes either form orthogonal to rounding ("twice is enough").  With the ~100
// eigenvalues the 1e-10 cut drops from a 900 x 900 Wa in ONE cluster this loop was 10 of the 10.8 ms of inverse iteration.
NLE_SIMD_CLONES static void orth_against_cluster(int n, const double* Zc, int m, double* x) {
    for (int pass = 0; pass < 2; ++pass) {
        int q = 0;
        for (; q + 4 <= m; q += 4) {
            const double *z0 = Zc + (size_t)q * n, *z1 = z0 + n, *z2 = z1 + n, *z3 = z2 + n;
            double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;
#pragma omp simd reduction(+ : h0, h1, h2, h3)
            for (int i = 0; i < n; ++i) {
                const double xi = x[i];
                h0 += z0[i] * xi;
                h1 += z1[i] * xi;
                h2 += z2[i] * xi;
                h3 += z3[i] * xi;
            }
#pragma omp simd
            for (int i = 0; i < n; ++i) x[i] -= (h0 * z0[i] + h1 * z1[i]) + (h2 * z2[i] + h3 * z3[i]);
        }
        for (; q < m; ++q) {
            const double* zq = Zc + (size_t)q * n;
            double h = 0.0;
#pragma omp simd reduction(+ : h)
            for (int i = 0; i < n; ++i) h += zq[i] * x[i];
#pragma omp simd
            for (int i = 0; i < n; ++i) x[i] -= h * zq[i];
        }
    }
}

// Inverse iteration (the scheme of tridiag_inverse_iteration below, operation for operation) for EIGHT isolated eigenvalues
// at once, one per AVX-512 lane: the LU factorisation with partial pivoting and the two substitutions are chains of n
// dependent steps, so eight shifts cost what one does.  Pivot choices differ per lane: both branches are formed and
// blended.  Same operations in the same order as the scalar code (whose compiler may contract or reorder a sum or two: the
// vectors agree to a few ulp, tools/micro/bisect_check.py).  lam8: the shifts (m <= 8 used, the rest repeat the last), idx8: their vector numbers (seeds), Zcols: where the
// vectors go.  false: a lane did not converge (the caller falls back exactly as for the scalar path).
#if defined(__x86_64__)
__attribute__((target("avx512f"))) static inline __m512d x8_fix_tiny(__m512d v, __m512d vtiny) {  // |v| < tiny: copysign(tiny, v == 0 ? 1 : v)
    const __m512d zero = _mm512_setzero_pd();
    const __m512i signbit = _mm512_set1_epi64((long long)0x8000000000000000ull);
    const __mmask8 small = _mm512_cmp_pd_mask(_mm512_abs_pd(v), vtiny, _CMP_LT_OQ);
    const __mmask8 iszero = _mm512_cmp_pd_mask(v, zero, _CMP_EQ_OQ);
    __m512i sg = _mm512_and_epi64(_mm512_castpd_si512(v), signbit);
    sg = _mm512_mask_mov_epi64(sg, iszero, _mm512_setzero_si512());
    const __m512d t = _mm512_castsi512_pd(_mm512_or_epi64(_mm512_castpd_si512(vtiny), sg));
    return _mm512_mask_mov_pd(v, small, t);
}
struct X8Arr {  // n vectors of eight doubles in 64-byte aligned storage (std::vector<__m512d> drops the alignment)
    double* p;
    __attribute__((target("avx512f"))) inline __m512d operator()(int i) const { return _mm512_load_pd(p + (size_t)8 * i); }
    __attribute__((target("avx512f"))) inline void set(int i, __m512d v) const { _mm512_store_pd(p + (size_t)8 * i, v); }
};
__attribute__((target("avx512f"))) static bool inverse_iteration_x8(int n, const double* d, const double* e, const double* lam8,
                                                                   const int* idx8, int m, double onenrm, double* const* Zcols) {
    const double eps = std::ldexp(1.0, -52), tiny = eps * onenrm;
    std::vector<double> store((size_t)7 * n * 8 + 8);
    double* base = store.data();
    while (reinterpret_cast<uintptr_t>(base) & 63) ++base;
    X8Arr a{base}, ra{base + (size_t)8 * n}, b{base + (size_t)16 * n}, c{base + (size_t)24 * n}, dd{base + (size_t)32 * n},
        x{base + (size_t)40 * n}, y{base + (size_t)48 * n};
    std::vector<__mmask8> piv(n);
    const __m512d vtiny = _mm512_set1_pd(tiny), zero = _mm512_setzero_pd();
    double lamv[8];
    for (int l = 0; l < 8; ++l) lamv[l] = lam8[l < m ? l : m - 1];
    const __m512d xj = _mm512_loadu_pd(lamv);
    for (int i = 0; i < n; ++i) {
        a.set(i, _mm512_sub_pd(_mm512_set1_pd(d[i]), xj));
        {
            const __m512d ev = _mm512_set1_pd(i + 1 < n ? e[i + 1] : 0.0);
            b.set(i, ev);
            c.set(i, ev);
        }
        dd.set(i, zero);
    }
    for (int i = 0; i + 1 < n; ++i) {
        const __m512d ai = a(i), ci = c(i), an = a(i + 1), bi = b(i);
        const __mmask8 ge = _mm512_cmp_pd_mask(_mm512_abs_pd(ai), _mm512_abs_pd(ci), _CMP_GE_OQ);
        const __m512d af = x8_fix_tiny(ai, vtiny);
        const __m512d num = _mm512_mask_blend_pd(ge, ai, ci), den = _mm512_mask_blend_pd(ge, ci, af);
        const __m512d mult = _mm512_div_pd(num, den);
        const __m512d X = _mm512_mask_blend_pd(ge, bi, an), Yv = _mm512_mask_blend_pd(ge, an, bi);
        a.set(i + 1, _mm512_sub_pd(X, _mm512_mul_pd(mult, Yv)));
        a.set(i, _mm512_mask_blend_pd(ge, ci, af));
        if (i + 2 < n) {
            const __m512d bn = b(i + 1);
            dd.set(i, _mm512_mask_blend_pd(ge, bn, zero));
            b.set(i + 1, _mm512_mask_blend_pd(ge, _mm512_mul_pd(_mm512_sub_pd(zero, mult), bn), bn));
        }
        b.set(i, _mm512_mask_blend_pd(ge, an, bi));
        c.set(i, mult);
        piv[i] = (__mmask8)~ge;
    }
    a.set(n - 1, x8_fix_tiny(a(n - 1), vtiny));
    const __m512d one = _mm512_set1_pd(1.0);
    for (int i = 0; i < n; ++i) ra.set(i, _mm512_div_pd(one, a(i)));
    {
        unsigned long long seed[8];
        for (int l = 0; l < 8; ++l) seed[l] = 0x2545F4914F6CDD1Dull ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(idx8[l < m ? l : m - 1] + 1));
        double xs[8];
        for (int i = 0; i < n; ++i) {
            for (int l = 0; l < 8; ++l) {
                seed[l] ^= seed[l] << 13, seed[l] ^= seed[l] >> 7, seed[l] ^= seed[l] << 17;
                xs[l] = (double)(seed[l] >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            }
            x.set(i, _mm512_loadu_pd(xs));
        }
    }
    const __m512d veps = _mm512_set1_pd(eps), vfloor = _mm512_set1_pd(1e-300), vn = _mm512_set1_pd((double)n * onenrm);
    const __m512d an1 = _mm512_max_pd(veps, _mm512_abs_pd(a(n - 1)));
    __mmask8 done = 0;
    const __mmask8 want = (__mmask8)((1u << m) - 1u);
    for (int it = 0; it < 8 && (done & want) != want; ++it) {
        __m512d s1 = zero;
        for (int i = 0; i < n; ++i) s1 = _mm512_add_pd(s1, _mm512_abs_pd(x(i)));
        const __m512d scl = _mm512_div_pd(_mm512_mul_pd(vn, an1), _mm512_max_pd(s1, vfloor));
        __m512d yi = _mm512_mul_pd(x(0), scl);
        for (int i = 0; i + 1 < n; ++i) {  // forward: row interchanges and multipliers
            const __m512d yn = _mm512_mul_pd(x(i + 1), scl);
            const __m512d p0 = _mm512_mask_blend_pd(piv[i], yi, yn), p1 = _mm512_mask_blend_pd(piv[i], yn, yi);
            y.set(i, p0);
            yi = _mm512_sub_pd(p1, _mm512_mul_pd(c(i), p0));
        }
        y.set(n - 1, yi);
        for (int i = n - 1; i >= 0; --i) {  // back substitution with the three upper diagonals
            __m512d t = y(i);
            if (i + 1 < n) t = _mm512_sub_pd(t, _mm512_mul_pd(b(i), x(i + 1)));
            if (i + 2 < n) t = _mm512_sub_pd(t, _mm512_mul_pd(dd(i), x(i + 2)));
            x.set(i, _mm512_mul_pd(t, ra(i)));
        }
        __m512d nrm = zero;
        for (int i = 0; i < n; ++i) nrm = _mm512_add_pd(nrm, _mm512_mul_pd(x(i), x(i)));
        nrm = _mm512_sqrt_pd(nrm);
        {
            double nv[8];
            _mm512_storeu_pd(nv, nrm);
            for (int l = 0; l < m; ++l)
                if (!(nv[l] > 0.0) || !std::isfinite(nv[l])) return false;
        }
        for (int i = 0; i < n; ++i) x.set(i, _mm512_div_pd(x(i), nrm));
        if (it >= 1) {  // residual ||(T - lam I) x||_2 against the ORIGINAL eigenvalue
            __m512d res = zero;
            for (int i = 0; i < n; ++i) {
                __m512d t = _mm512_mul_pd(_mm512_sub_pd(_mm512_set1_pd(d[i]), xj), x(i));
                if (i > 0) t = _mm512_add_pd(t, _mm512_mul_pd(_mm512_set1_pd(e[i]), x(i - 1)));
                if (i + 1 < n) t = _mm512_add_pd(t, _mm512_mul_pd(_mm512_set1_pd(e[i + 1]), x(i + 1)));
                res = _mm512_add_pd(res, _mm512_mul_pd(t, t));
            }
            const __mmask8 ok = _mm512_cmp_pd_mask(_mm512_sqrt_pd(res), _mm512_set1_pd(1e3 * eps * onenrm), _CMP_LE_OQ);
            const __mmask8 fresh = (__mmask8)(ok & ~done & want);
            if (fresh) {
                double xs[8];
                for (int i = 0; i < n; ++i) {
                    _mm512_storeu_pd(xs, x(i));
                    for (int l = 0; l < m; ++l)
                        if (fresh & (1u << l)) Zcols[l][i] = xs[l];
                }
                done |= fresh;
            }
        }
    }
    return (done & want) == want;
}
#endif

// Eigenvectors of the symmetric tridiagonal T (diagonal d[0..n), sub-diagonal e[1..n)) for k of its eigenvalues
// lam[0..k), given in DESCENDING order and accurate to rounding, by inverse iteration -- the scheme of LAPACK's dstein:
// LU of T - lam I with partial pivoting (tiny pivots perturbed), a few solves from a pseudo-random start, vectors of
// eigenvalues closer than 1e-3 ||T|| re-orthogonalised against each other (modified Gram-Schmidt, twice).  O(n k)
// instead of the O(n^2 k)-ish rotation sweeps: what sym_eigen_top uses when only the leading part of the spectrum is
// wanted (orthogonalize keeps K of q eigenvectors, src/filter.cpp:314).  Z: n x k column-major.  false: a vector did not
// reach a residual of 1e3 eps ||T|| (the caller falls back to the rotation form).
bool tridiag_inverse_iteration(int n, const double* d, const double* e, const double* lam, int k, double* Z) {
    const double eps = std::ldexp(1.0, -52);
    double onenrm = 0.0;
    for (int i = 0; i < n; ++i)
        onenrm = std::max(onenrm, std::fabs(d[i]) + (i > 0 ? std::fabs(e[i]) : 0.0) + (i + 1 < n ? std::fabs(e[i + 1]) : 0.0));
    if (onenrm == 0.0) onenrm = 1.0;
    const double ortol = 1e-3 * onenrm, tiny = eps * onenrm;
    // Clusters (eigenvalues closer than ortol to their neighbour; their vectors are re-orthogonalised against each other)
    // are independent of one another: the shift perturbation below never reaches across a gap > ortol, and every vector
    // starts from its own pseudo-random vector (seeded by its index, so the result does not depend on who computes it).
    // With many vectors the clusters are dealt to a few threads.
    std::vector<int> starts;
    for (int j = 0; j < k; ++j)
        if (j == 0 || std::fabs(lam[j] - lam[j - 1]) > ortol) starts.push_back(j);
    starts.push_back(k);
    const int ngroups = (int)starts.size() - 1;
    auto do_group = [&](int gp, int gend) -> bool {
    std::vector<double> a(n), ra(n), b(n), c(n), dd(n), x(n), y(n);
    std::vector<char> piv(n);
    double prev = 0.0;
    for (int j = gp; j < gend; ++j) {
        unsigned long long seed = 0x2545F4914F6CDD1Dull ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(j + 1));
        double xj = lam[j];
        if (j > gp) {
            const double pertol = 10.0 * eps * std::max(std::fabs(xj), tiny);
            if (prev - xj < pertol) xj = prev - pertol;  // keep the shifts distinct (descending order)
        }
        prev = xj;
        // LU of T - xj I with partial pivoting: row k holds (a, b, dd), multipliers in c
        for (int i = 0; i < n; ++i) {
            a[i] = d[i] - xj;
            b[i] = (i + 1 < n) ? e[i + 1] : 0.0;
            c[i] = (i + 1 < n) ? e[i + 1] : 0.0;
            dd[i] = 0.0;
        }
        for (int i = 0; i + 1 < n; ++i) {
            if (std::fabs(a[i]) >= std::fabs(c[i])) {
                if (std::fabs(a[i]) < tiny) a[i] = std::copysign(tiny, a[i] == 0.0 ? 1.0 : a[i]);
                const double mult = c[i] / a[i];
                a[i + 1] -= mult * b[i];
                c[i] = mult;
                piv[i] = 0;
            } else {
                const double mult = a[i] / c[i];
                a[i] = c[i];
                const double t = a[i + 1];
                a[i + 1] = b[i] - mult * t;
                if (i + 2 < n) {
                    dd[i] = b[i + 1];
                    b[i + 1] = -mult * dd[i];
                }
                b[i] = t;
                c[i] = mult;
                piv[i] = 1;
            }
        }
        if (std::fabs(a[n - 1]) < tiny) a[n - 1] = std::copysign(tiny, a[n - 1] == 0.0 ? 1.0 : a[n - 1]);
        for (int i = 0; i < n; ++i) ra[i] = 1.0 / a[i];  // the back substitution is a dependent chain: multiply, not divide
        for (int i = 0; i < n; ++i) {
            seed ^= seed << 13, seed ^= seed >> 7, seed ^= seed << 17;
            x[i] = (double)(seed >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        }
        double* z = Z + (size_t)j * n;
        bool ok = false;
        for (int it = 0; it < 8 && !ok; ++it) {
            // scale the right-hand side so that the solve cannot overflow (dstein: ||x||_1 = n ||T|| max(eps, |u_nn|))
            double s1 = 0.0;
            for (int i = 0; i < n; ++i) s1 += std::fabs(x[i]);
            const double scl = n * onenrm * std::max(eps, std::fabs(a[n - 1])) / std::max(s1, 1e-300);
            for (int i = 0; i < n; ++i) y[i] = x[i] * scl;
            for (int i = 0; i + 1 < n; ++i) {  // forward: apply the row interchanges and multipliers
                if (piv[i]) std::swap(y[i], y[i + 1]);
                y[i + 1] -= c[i] * y[i];
            }
            for (int i = n - 1; i >= 0; --i) {  // back substitution with the three upper diagonals
                double t = y[i];
                if (i + 1 < n) t -= b[i] * x[i + 1];
                if (i + 2 < n) t -= dd[i] * x[i + 2];
                x[i] = t * ra[i];
            }
            orth_against_cluster(n, Z + (size_t)gp * n, j - gp, x.data());  // the cluster's earlier vectors, twice
            double nrm = 0.0;
            for (int i = 0; i < n; ++i) nrm += x[i] * x[i];
            nrm = std::sqrt(nrm);
            if (!(nrm > 0.0) || !std::isfinite(nrm)) return false;
            for (int i = 0; i < n; ++i) x[i] /= nrm;
            if (it >= 1) {  // residual ||(T - lam I) x||_2 against the ORIGINAL eigenvalue
                double res = 0.0;
                for (int i = 0; i < n; ++i) {
                    double t = (d[i] - lam[j]) * x[i];
                    if (i > 0) t += e[i] * x[i - 1];
                    if (i + 1 < n) t += e[i + 1] * x[i + 1];
                    res += t * t;
                }
                ok = std::sqrt(res) <= 1e3 * eps * onenrm;
            }
        }
        if (!ok) return false;
        std::copy(x.begin(), x.end(), z);
    }
    return true;
    };
    // A LARGE cluster (the ~100 eigenvalues the 1e-10 cut drops from a 900 x 900 Wa, all within ortol of each other) spends
    // its time in the vector-by-vector re-orthogonalisation, a chain nothing can be overlapped with.  Its eigenvalues are
    // known to ~eps ||T|| while they differ by far more, so plain inverse iteration (no orthogonalisation, every vector
    // independent of the others: threads) already isolates each vector to ~eps ||T|| / gap; ONE Cholesky-QR of the block
    // (two level-3 products) then makes it orthonormal to rounding, moving every vector by that same small amount inside
    // the cluster's invariant subspace.  Numerically repeated eigenvalues give (random, independent) vectors of their
    // eigenspace -- a worse conditioned block: then a second Cholesky-QR, and if the Gram matrix does not factor at all the
    // cluster goes through the sequential path.  Residuals are checked against the same bound as there.
    constexpr int kBlockMin = 24;
    auto do_group_block = [&](int gp, int gend) -> bool {
        const int mc = gend - gp;
        std::vector<double> shift(mc);
        {
            double prev = 0.0;
            for (int j = gp; j < gend; ++j) {
                double xj = lam[j];
                if (j > gp) {
                    const double pertol = 10.0 * eps * std::max(std::fabs(xj), tiny);
                    if (prev - xj < pertol) xj = prev - pertol;
                }
                prev = xj;
                shift[j - gp] = xj;
            }
        }
        double* Zg = Z + (size_t)gp * n;
        const int nt = std::min(4, std::max(1, mc / 8));
        std::vector<char> okv(nt, 1);
        run_split(nt, nt, [&](int t) {
            std::vector<double> a(n), ra(n), b(n), c(n), dd(n), x(n), y(n);
            std::vector<char> piv(n);
            for (int jj = (int)((long long)mc * t / nt); jj < (int)((long long)mc * (t + 1) / nt); ++jj) {
                const int j = gp + jj;
                const double xj = shift[jj];
                unsigned long long seed = 0x2545F4914F6CDD1Dull ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(j + 1));
                for (int i = 0; i < n; ++i) {
                    a[i] = d[i] - xj;
                    b[i] = (i + 1 < n) ? e[i + 1] : 0.0;
                    c[i] = (i + 1 < n) ? e[i + 1] : 0.0;
                    dd[i] = 0.0;
                }
                for (int i = 0; i + 1 < n; ++i) {
                    if (std::fabs(a[i]) >= std::fabs(c[i])) {
                        if (std::fabs(a[i]) < tiny) a[i] = std::copysign(tiny, a[i] == 0.0 ? 1.0 : a[i]);
                        const double mult = c[i] / a[i];
                        a[i + 1] -= mult * b[i];
                        c[i] = mult;
                        piv[i] = 0;
                    } else {
                        const double mult = a[i] / c[i];
                        a[i] = c[i];
                        const double tt = a[i + 1];
                        a[i + 1] = b[i] - mult * tt;
                        if (i + 2 < n) {
                            dd[i] = b[i + 1];
                            b[i + 1] = -mult * dd[i];
                        }
                        b[i] = tt;
                        c[i] = mult;
                        piv[i] = 1;
                    }
                }
                if (std::fabs(a[n - 1]) < tiny) a[n - 1] = std::copysign(tiny, a[n - 1] == 0.0 ? 1.0 : a[n - 1]);
                for (int i = 0; i < n; ++i) ra[i] = 1.0 / a[i];
                for (int i = 0; i < n; ++i) {
                    seed ^= seed << 13, seed ^= seed >> 7, seed ^= seed << 17;
                    x[i] = (double)(seed >> 11) * (1.0 / 9007199254740992.0) - 0.5;
                }
                for (int it = 0; it < 3; ++it) {
                    double s1 = 0.0;
                    for (int i = 0; i < n; ++i) s1 += std::fabs(x[i]);
                    const double scl = n * onenrm * std::max(eps, std::fabs(a[n - 1])) / std::max(s1, 1e-300);
                    for (int i = 0; i < n; ++i) y[i] = x[i] * scl;
                    for (int i = 0; i + 1 < n; ++i) {
                        if (piv[i]) std::swap(y[i], y[i + 1]);
                        y[i + 1] -= c[i] * y[i];
                    }
                    for (int i = n - 1; i >= 0; --i) {
                        double tt = y[i];
                        if (i + 1 < n) tt -= b[i] * x[i + 1];
                        if (i + 2 < n) tt -= dd[i] * x[i + 2];
                        x[i] = tt * ra[i];
                    }
                    double nrm = 0.0;
                    for (int i = 0; i < n; ++i) nrm += x[i] * x[i];
                    nrm = std::sqrt(nrm);
                    if (!(nrm > 0.0) || !std::isfinite(nrm)) {
                        okv[t] = 0;
                        return;
                    }
                    for (int i = 0; i < n; ++i) x[i] /= nrm;
                }
                std::copy(x.begin(), x.end(), Zg + (size_t)jj * n);
            }
        });
        for (char v : okv)
            if (!v) return false;
        // Cholesky-QR: G = Z^T Z = L L^T, Z <- Z L^-T; again if the block was far from orthonormal
        std::vector<double> Zt((size_t)mc * n), G((size_t)mc * mc), Li((size_t)mc * mc), Zn((size_t)n * mc);
        for (int pass = 0; pass < 2; ++pass) {
            for (int j = 0; j < mc; ++j)
                for (int i = 0; i < n; ++i) Zt[(size_t)i * mc + j] = Zg[(size_t)j * n + i];
            gemm_nn_cols(Zt.data(), Zg, G.data(), mc, n, mc, 0, mc);  // (mc x n) (n x mc)
            double off = 0.0;
            for (int j = 0; j < mc; ++j)
                for (int i = 0; i < mc; ++i)
                    if (i != j) off = std::max(off, std::fabs(G[(size_t)j * mc + i]));
            if (pass == 1 && off < 1e-13) break;  // already orthonormal to rounding
            for (int j = 0; j < mc; ++j)
                for (int i = 0; i < j; ++i) G[(size_t)j * mc + i] = 0.0;  // lower triangle only
            if (!cholesky_lower(mc, G.data(), 1e-10)) return false;  // (nearly) dependent vectors: the sequential path
            lower_inverse(mc, G.data(), Li.data());
            gemm_nt_cols(Zg, Li.data(), Zn.data(), n, mc, mc, 0, mc);  // Z L^-T
            std::copy(Zn.begin(), Zn.end(), Zg);
            if (pass == 0 && off < 1e-5) break;  // one pass leaves an error ~ eps (1 + off^2 ...): enough
        }
        for (int jj = 0; jj < mc; ++jj) {  // residual ||(T - lam I) x||_2 against the ORIGINAL eigenvalue
            const double* x = Zg + (size_t)jj * n;
            const double lj = lam[gp + jj];
            double res = 0.0;
            for (int i = 0; i < n; ++i) {
                double tt = (d[i] - lj) * x[i];
                if (i > 0) tt += e[i] * x[i - 1];
                if (i + 1 < n) tt += e[i + 1] * x[i + 1];
                res += tt * tt;
            }
            if (!(std::sqrt(res) <= 1e3 * eps * onenrm)) return false;
        }
        return true;
    };
    static const bool no_block = std::getenv("NLE_EIG_NO_BLOCK") != nullptr;
    std::vector<char> done_group(ngroups, 0);
    if (!no_block)
        for (int g = 0; g < ngroups; ++g)
            if (starts[g + 1] - starts[g] >= kBlockMin) {
                const bool blk = do_group_block(starts[g], starts[g + 1]);
                if (std::getenv("NLE_EIG_TRACE"))
                    std::fprintf(stderr, "[nle eig] cluster of %d vectors: %s\n", starts[g + 1] - starts[g],
                                 blk ? "block inverse iteration + Cholesky-QR" : "block form gave up, vector by vector");
                if (!blk && !do_group(starts[g], starts[g + 1])) return false;
                done_group[g] = 1;
            }
#if defined(__x86_64__)
    {   // isolated eigenvalues (clusters of one: nothing to orthogonalise against), eight per sweep
        static const bool x8 = __builtin_cpu_supports("avx512f") && std::getenv("NLE_EIG_NO_X8") == nullptr;
        std::vector<int> single;
        if (x8 && n >= 16)
            for (int g = 0; g < ngroups; ++g)
                if (!done_group[g] && starts[g + 1] - starts[g] == 1) single.push_back(g);
        if (single.size() >= 3)
            for (size_t s0 = 0; s0 < single.size(); s0 += 8) {
                const int m8 = (int)std::min<size_t>(8, single.size() - s0);
                double lam8[8];
                int idx8[8];
                double* zc[8];
                for (int l = 0; l < m8; ++l) {
                    const int j = starts[single[s0 + l]];
                    lam8[l] = lam[j];
                    idx8[l] = j;
                    zc[l] = Z + (size_t)j * n;
                }
                if (!inverse_iteration_x8(n, d, e, lam8, idx8, m8, onenrm, zc)) return false;
                for (int l = 0; l < m8; ++l) done_group[single[s0 + l]] = 1;
            }
    }
#endif
    // work of a cluster of m vectors ~ m n (8 + m) (solves + re-orthogonalisation), ~2.6 ns per unit on the GPU box's cores;
    // threads (~0.1 ms to start and join) pay off from ~0.8 ms of it
    double work = 0.0;
    for (int g = 0; g < ngroups; ++g) {
        const double m = done_group[g] ? 0.0 : starts[g + 1] - starts[g];
        work += m * n * (8.0 + m);
    }
    int nthreads = (ngroups >= 2 && work > 3e5) ? std::min(4, ngroups) : 1;
    if (const char* ev = std::getenv("NLE_EIG_THREADS")) nthreads = std::max(1, std::min(std::atoi(ev), ngroups));
    if (nthreads <= 1) {
        for (int g = 0; g < ngroups; ++g)
            if (!done_group[g] && !do_group(starts[g], starts[g + 1])) return false;
        return true;
    }
    // contiguous runs of clusters of about equal work per thread
    std::vector<int> cut(nthreads + 1, ngroups);
    cut[0] = 0;
    {
        double acc = 0.0;
        int t = 1;
        for (int g = 0; g < ngroups && t < nthreads; ++g) {
            const double m = done_group[g] ? 0.0 : starts[g + 1] - starts[g];
            acc += m * n * (8.0 + m);
            if (acc >= work * t / nthreads) cut[t++] = g + 1;
        }
    }
    std::vector<char> okv(nthreads, 1);
    run_split(nthreads, nthreads, [&](int t) {
        for (int g = cut[t]; g < cut[t + 1]; ++g)
            if (!done_group[g] && !do_group(starts[g], starts[g + 1])) {
                okv[t] = 0;
                return;
            }
    });
    for (char v : okv)
        if (!v) return false;
    return true;
}

bool tridiag_eigenvectors(int n, const double* d, const double* e, const double* lam_all, int first, int count, double* Z) {
    if (count <= 0) return true;
    if (tridiag_inverse_iteration(n, d, e, lam_all + first, count, Z)) return true;
    // the classic iteration on T itself: rotations accumulated from the identity
    std::vector<double> V((size_t)n * n, 0.0), dd(d, d + n), ee(e, e + n);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    if (!ql_implicit(n, V.data(), dd.data(), ee.data())) return false;
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return dd[a] < dd[b]; });
    std::reverse(idx.begin(), idx.end());
    for (int j = 0; j < count; ++j)
        std::copy(V.begin() + (size_t)idx[first + j] * n, V.begin() + (size_t)idx[first + j] * n + n, Z + (size_t)j * n);
    return true;
}

// The part of sym_eigen_top after the reduction: V (n x n, u_i in column i rows 0..i-1), hs, and the tridiagonal (d, e) as
// tridiag_reduce -- or the device kernel k_tridiag (tridiag.hip), same conventions -- leaves them.
bool sym_eigen_top_reduced(int n, const double* V, const double* d_in, const double* e_in, const double* hs, int ncols,
                           int nthreads, double* U, double* D) {
    if (nthreads <= 0) nthreads = default_threads(n, ncols);
    static const bool trace = std::getenv("NLE_EIG_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0.0;
    std::vector<double> d(d_in, d_in + n), e(e_in, e_in + n);
    const std::vector<double> d0(d), e0(e);  // T itself (the QL iteration overwrites d and e)
    std::vector<Sweep> sweeps;
    std::vector<double> cs, sn;
    // a leading part of the spectrum: the eigenvectors come from inverse iteration, the rotations are not needed (should
    // the inverse iteration give up, the iteration is repeated with recording below)
    const bool invit = ncols > 0 && 2 * ncols <= n && std::getenv("NLE_EIG_NO_INVIT") == nullptr;
    if (invit || ncols == 0) {
        if (!ql_values(n, d.data(), e.data())) return false;
    } else {
        cs.reserve((size_t)n * n);
        sn.reserve((size_t)n * n);
        if (!ql_record(n, d.data(), e.data(), sweeps, cs, sn)) return false;
    }
    const double t1 = trace ? now() : 0.0;
    // descending order (stable on the ascending sort the classic path uses, reversed)
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
    std::reverse(idx.begin(), idx.end());
    for (int j = 0; j < n; ++j) D[j] = d[idx[j]];
    if (ncols == 0) return true;
    if (invit) {
        // a leading part of the spectrum only: inverse iteration on T for those eigenvalues, then the back-transformation
        if (tridiag_inverse_iteration(n, d0.data(), e0.data(), D, ncols, U)) {
            const double t2 = trace ? now() : 0.0;
            const int cparts0 = std::max(1, std::min(nthreads, (ncols + 3) / 4));
            run_split(cparts0, nthreads, [&](int q) {
                const int j0 = (int)((long long)ncols * q / cparts0), j1 = (int)((long long)ncols * (q + 1) / cparts0);
                back_transform_cols(n, V, hs, U, j0, j1);
            });
            if (trace)
                std::fprintf(stderr, "[nle eig] n = %d, %d vectors: QL %.3f ms, inverse iteration %.3f ms, back-transformation %.3f ms\n",
                             n, ncols, t1 - t0, t2 - t1, now() - t2);
            return true;
        }
        // inverse iteration gave up: the rotations after all (same iteration, same eigenvalues)
        d = d0;
        e = e0;
        cs.reserve((size_t)n * n);
        sn.reserve((size_t)n * n);
        if (!ql_record(n, d.data(), e.data(), sweeps, cs, sn)) return false;
    }
    const int ldz = (n + 7) & ~7;  // rows padded to whole 8-row vectors (the padding stays zero)
    // (64-byte aligned: the 64-row blocks of different threads then never share a cache line)
    std::vector<double> Zbuf((size_t)ldz * n + 8, 0.0);
    double* Z = Zbuf.data();
    while (reinterpret_cast<uintptr_t>(Z) & 63) ++Z;
    for (int i = 0; i < n; ++i) Z[(size_t)i * ldz + i] = 1.0;
    const int nvec = ldz / 8, nblocks = (nvec + 7) / 8;  // 64-row blocks
    run_split(nblocks, nthreads, [&](int b) {
        apply_rotations_rows(ldz, Z, b * 64, std::min(8, nvec - b * 8), sweeps.data(), sweeps.size(), cs.data(), sn.data());
    });
    for (int j = 0; j < ncols; ++j) std::copy(Z + (size_t)idx[j] * ldz, Z + (size_t)idx[j] * ldz + n, U + (size_t)j * n);
    const int cparts = std::max(1, std::min(nthreads, (ncols + 3) / 4));
    run_split(cparts, nthreads, [&](int q) {
        const int j0 = (int)((long long)ncols * q / cparts), j1 = (int)((long long)ncols * (q + 1) / cparts);
        back_transform_cols(n, V, hs, U, j0, j1);
    });
    return true;
}

bool sym_eigen_top(const double* M, int n, int ncols, int nthreads, double* U, double* D) {
    if (n <= 0) return true;
    ncols = std::max(0, std::min(ncols, n));
    if (n == 1) {
        D[0] = M[0];
        if (ncols) U[0] = 1.0;
        return true;
    }
    std::vector<double> V((size_t)n * n), d(n), e(n), hs(n);
    for (int c = 0; c < n; ++c)  // mirror the lower triangle (SelfAdjointEigenSolver reads only the lower one)
        for (int r = 0; r < n; ++r) V[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : M[(size_t)r * n + c];
    tridiag_reduce(n, V.data(), d.data(), e.data(), hs.data());
    return sym_eigen_top_reduced(n, V.data(), d.data(), e.data(), hs.data(), ncols, nthreads, U, D);
}

// eigen_decomposition_top for a matrix already reduced (n >= 2, kmax <= n): all eigenvalues descending in D, the first kmax
// eigenvectors in U, *r_out = length of the leading run >= eps (reference src/filter.cpp:209-216)
bool eigen_decomposition_top_reduced(int n, double eps, int kmax, const double* V, const double* d, const double* e,
                                     const double* hs, double* U, double* D, int* r_out) {
    kmax = std::max(0, std::min(kmax, n));
    if (!sym_eigen_top_reduced(n, V, d, e, hs, kmax, 0, U, D)) return false;
    int r = 0;
    while (r < n && D[r] >= eps) ++r;
    *r_out = r;
    return true;
}

// ---- a FEW eigenvalues of the tridiagonal form by bisection on Sturm counts, up to 64 shifts per sweep of T.
// What the train path needs of Q is its K leading eigenpairs and the NUMBER of eigenvalues >= 1e-10 (src/filter.cpp:313-316), of
// Wa the number below 1e-10, those few eigenvalues, and the largest (the deflated root): an implicit QL run for all n
// eigenvalues (0.31 ms at n = 200 on the GPU box's host, a chain of dependent rotations) is replaced by counts.  A count is
// a chain of n - 1 dependent divisions (~20 cycles each): eight shifts ride in one AVX-512 vector and up to eight vectors are
// interleaved so the divider stays busy -- 50 eigenvalues are ONE sweep of T per halving, ~55 halvings.  Accuracy: the
// count is exact for a tridiagonal matrix within a few ulp ||T|| of T (dstebz's recurrence with its pivmin guard), the same
// backward error the QL iteration has; the rank cut at eps is one more count.
namespace {
typedef long long v8l __attribute__((vector_size(64)));
struct SturmT {
    int n;
    const double* d;
    std::vector<double> e2;  // e2[i] = e[i]^2 couples i - 1 and i (e2[0] = 0)
    double pivmin, lo, hi;   // Gershgorin bounds, widened
};
SturmT sturm_setup(int n, const double* d, const double* e) {
    SturmT t{n, d, std::vector<double>(n, 0.0), 0.0, 0.0, 0.0};
    double emax2 = 0.0, lo = d[0], hi = d[0];
    for (int i = 1; i < n; ++i) {
        t.e2[i] = e[i] * e[i];
        emax2 = std::max(emax2, t.e2[i]);
    }
    for (int i = 0; i < n; ++i) {
        const double r = (i > 0 ? std::fabs(e[i]) : 0.0) + (i + 1 < n ? std::fabs(e[i + 1]) : 0.0);
        lo = std::min(lo, d[i] - r);
        hi = std::max(hi, d[i] + r);
    }
    const double nrm = std::max(std::fabs(lo), std::fabs(hi));
    t.pivmin = 2.2250738585072014e-308 * std::max(1.0, emax2);
    t.lo = lo - 2.0 * 2.220446049250313e-16 * nrm * n - 2.0 * t.pivmin;
    t.hi = hi + 2.0 * 2.220446049250313e-16 * nrm * n + 2.0 * t.pivmin;
    return t;
}
// cnt[v] = number of eigenvalues of T smaller than each of the shifts s[v] (NV vectors of eight).  Intrinsics, not vector
// extensions: GCC scalarises 512-bit vector comparisons inside target_clones (measured: 48 cycles per vector step).
template <int NV>
__attribute__((target("avx512f"))) static void sturm_counts_avx512(int n, const double* d, const double* e2, double pivmin,
                                                                    const v8d* s, v8l* cnt) {
    const __m512d pm = _mm512_set1_pd(pivmin), npm = _mm512_set1_pd(-pivmin), zero = _mm512_setzero_pd();
    const __m512i one = _mm512_set1_epi64(1);
    __m512d q[NV], sv[NV];
    __m512i c[NV];
    for (int v = 0; v < NV; ++v) {
        sv[v] = _mm512_loadu_pd(&s[v]);
        q[v] = _mm512_sub_pd(_mm512_set1_pd(d[0]), sv[v]);
        q[v] = _mm512_mask_mov_pd(q[v], _mm512_cmp_pd_mask(_mm512_abs_pd(q[v]), pm, _CMP_LT_OQ), npm);
        c[v] = _mm512_maskz_mov_epi64(_mm512_cmp_pd_mask(q[v], zero, _CMP_LT_OQ), one);
    }
    for (int i = 1; i < n; ++i) {
        const __m512d di = _mm512_set1_pd(d[i]), ei = _mm512_set1_pd(e2[i]);
        for (int v = 0; v < NV; ++v) {
            q[v] = _mm512_sub_pd(_mm512_sub_pd(di, sv[v]), _mm512_div_pd(ei, q[v]));
            q[v] = _mm512_mask_mov_pd(q[v], _mm512_cmp_pd_mask(_mm512_abs_pd(q[v]), pm, _CMP_LT_OQ), npm);
            c[v] = _mm512_mask_add_epi64(c[v], _mm512_cmp_pd_mask(q[v], zero, _CMP_LT_OQ), c[v], one);
        }
    }
    for (int v = 0; v < NV; ++v) _mm512_storeu_si512(&cnt[v], c[v]);
}
void sturm_counts_scalar(int n, const double* d, const double* e2, double pivmin, const v8d* s, int nv, v8l* cnt) {
    for (int v = 0; v < nv; ++v) {
        double q[8], sh[8];
        long long c[8];
        for (int l = 0; l < 8; ++l) {
            sh[l] = s[v][l];
            q[l] = d[0] - sh[l];
            if (std::fabs(q[l]) < pivmin) q[l] = -pivmin;
            c[l] = q[l] < 0.0;
        }
        for (int i = 1; i < n; ++i)
            for (int l = 0; l < 8; ++l) {
                q[l] = (d[i] - sh[l]) - e2[i] / q[l];
                if (std::fabs(q[l]) < pivmin) q[l] = -pivmin;
                c[l] += q[l] < 0.0;
            }
        for (int l = 0; l < 8; ++l) cnt[v][l] = c[l];
    }
}
void sturm_counts(const SturmT& t, const v8d* s, int nv, v8l* cnt) {
    static const bool avx512 = __builtin_cpu_supports("avx512f");
    if (!avx512) return sturm_counts_scalar(t.n, t.d, t.e2.data(), t.pivmin, s, nv, cnt);
    switch (nv) {
        case 1: sturm_counts_avx512<1>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        case 2: sturm_counts_avx512<2>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        case 3: sturm_counts_avx512<3>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        case 4: sturm_counts_avx512<4>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        case 5: sturm_counts_avx512<5>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        case 6: sturm_counts_avx512<6>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        case 7: sturm_counts_avx512<7>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
        default: sturm_counts_avx512<8>(t.n, t.d, t.e2.data(), t.pivmin, s, cnt); break;
    }
}
int sturm_count1(const SturmT& t, double s) {
    const v8d sv = {s, s, s, s, s, s, s, s};
    v8l c;
    sturm_counts(t, &sv, 1, &c);
    return (int)c[0];
}
// the eigenvalues number k[0 .. m) of T counted from the SMALLEST (k = 0), m <= 64, to the last bit the counts resolve
void sturm_bisect(const SturmT& t, const int* k, int m, double* out) {
    const int nv = (m + 7) / 8;
    v8d lo[8], hi[8], mid[8];
    v8l kk[8], c[8];
    for (int l = 0; l < 8 * nv; ++l) {
        lo[l / 8][l % 8] = t.lo;
        hi[l / 8][l % 8] = t.hi;
        kk[l / 8][l % 8] = k[l < m ? l : m - 1];
    }
    // to one ulp of ||T|| (what any eigenvalue of T is defined to; a relative criterion would spend 40 more halvings on the
    // eigenvalues near the 1e-10 cut for digits the reduction never had)
    const double tol = 2.220446049250313e-16 * std::max(std::fabs(t.lo), std::fabs(t.hi)) + 2.0 * t.pivmin;
    for (int it = 0; it < 1100; ++it) {  // a double has 2098 binades at most; ~55 halvings when lambda is O(||T||)
        for (int v = 0; v < nv; ++v) mid[v] = 0.5 * (lo[v] + hi[v]);
        sturm_counts(t, mid, nv, c);
        bool done = true;
        for (int v = 0; v < nv; ++v) {
            const v8l up = c[v] > kk[v];  // more than k eigenvalues below mid: lambda_k < mid
            hi[v] = up ? mid[v] : hi[v];
            lo[v] = up ? lo[v] : mid[v];
            for (int l = 0; l < 8; ++l) {
                const double m2 = 0.5 * (lo[v][l] + hi[v][l]);
                if (hi[v][l] - lo[v][l] > tol && m2 > lo[v][l] && m2 < hi[v][l]) done = false;
            }
        }
        if (done) break;
    }
    for (int l = 0; l < m; ++l) out[l] = 0.5 * (lo[l / 8][l % 8] + hi[l / 8][l % 8]);
}
// eigenvalues with DESCENDING numbers [first, first + count) (number 0 = the largest) into out[0 .. count)
void sturm_eigenvalues_desc(const SturmT& t, int first, int count, double* out) {
    for (int j0 = 0; j0 < count; j0 += 64) {
        const int m = std::min(64, count - j0);
        int k[64];
        for (int l = 0; l < m; ++l) k[l] = t.n - 1 - (first + j0 + l);
        sturm_bisect(t, k, m, out + j0);
    }
}
}  // namespace

// All eigenvalues (DESCENDING, in D) and the eigenvectors of D[first .. first + count) only (U: n x count): the reduction
// without the orthogonal factor, QL on (d, e) for the values, inverse iteration on T for the selected vectors, and their
// back-transformation.  What the deflated root of Wa needs: the few eigenpairs the 1e-10 cut removes (pipeline.hip).
bool sym_eigen_select(const double* M, int n, double* D, int first, int count, double* U, double below_eps, int max_below,
                      int* kept_out) {
    if (n <= 0) return true;
    if (n == 1) {
        D[0] = M[0];
        if (count > 0) U[0] = 1.0;
        return true;
    }
    std::vector<double> V((size_t)n * n), d(n), e(n), hs(n);
    for (int c = 0; c < n; ++c)  // mirror the lower triangle (SelfAdjointEigenSolver reads only the lower one)
        for (int r = 0; r < n; ++r) V[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : M[(size_t)r * n + c];
    tridiag_reduce(n, V.data(), d.data(), e.data(), hs.data());
    const std::vector<double> d0(d), e0(e);
    if (!ql_values(n, d.data(), e.data())) return false;
    std::sort(d.begin(), d.end(), [](double a, double b) { return a > b; });
    std::copy(d.begin(), d.end(), D);
    if (kept_out) {  // select the eigenvalues after the leading run >= below_eps, if there are at most max_below of them
        int kept = 0;
        while (kept < n && D[kept] >= below_eps) ++kept;
        *kept_out = kept;
        first = kept;
        count = (n - kept <= max_below) ? n - kept : 0;
    }
    first = std::max(0, std::min(first, n));
    count = std::max(0, std::min(count, n - first));
    if (count == 0) return true;
    if (!tridiag_inverse_iteration(n, d0.data(), e0.data(), D + first, count, U)) return false;
    back_transform_cols(n, V.data(), hs.data(), U, 0, count);
    return true;
}

// What the deflated root of Wa needs and no more (ortho.hip): *kept_out = the number of eigenvalues >= eps (one Sturm
// count); if between 1 and max_below fall below it, those eigenvalues DESCENDING in Dbelow and their eigenvectors in U
// (n x max_below), the largest eigenvalue in *lam_max and the smallest kept one in *lam_min_kept, all by bisection.
bool sym_eigen_below(const double* M, int n, double eps, int max_below, int* kept_out, double* lam_max, double* lam_min_kept,
                     double* Dbelow, double* U) {
    if (n < 8 || std::getenv("NLE_EIG_NO_BISECT") != nullptr) {
        std::vector<double> D(n);
        if (!sym_eigen_select(M, n, D.data(), 0, 0, U, eps, max_below, kept_out)) return false;
        const int kept = *kept_out;
        *lam_max = D[0];
        *lam_min_kept = kept > 0 ? D[kept - 1] : 0.0;
        if (n - kept <= max_below) std::copy(D.begin() + kept, D.end(), Dbelow);
        return true;
    }
    std::vector<double> V((size_t)n * n), d(n), e(n), hs(n);
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) V[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : M[(size_t)r * n + c];
    tridiag_reduce(n, V.data(), d.data(), e.data(), hs.data());
    const SturmT t = sturm_setup(n, d.data(), e.data());
    const int kept = n - sturm_count1(t, eps);
    *kept_out = kept;
    const int nb = n - kept;
    {   // lambda_max, the smallest kept, and (if few) the ones below the cut: one or two vectors of shifts
        std::vector<int> k;
        k.push_back(n - 1);
        k.push_back(kept > 0 ? n - kept : n - 1);
        if (nb <= max_below)
            for (int j = 0; j < nb; ++j) k.push_back(nb - 1 - j);
        std::vector<double> out(k.size());
        for (size_t j0 = 0; j0 < k.size(); j0 += 64) sturm_bisect(t, k.data() + j0, (int)std::min<size_t>(64, k.size() - j0), out.data() + j0);
        *lam_max = out[0];
        *lam_min_kept = kept > 0 ? out[1] : 0.0;
        if (nb <= max_below) std::copy(out.begin() + 2, out.end(), Dbelow);
    }
    if (nb == 0 || nb > max_below) return true;
    if (!tridiag_inverse_iteration(n, d.data(), e.data(), Dbelow, nb, U)) return false;
    back_transform_cols(n, V.data(), hs.data(), U, 0, nb);
    return true;
}

// All eigenvectors on one thread: the classic reduction WITH accumulation of the orthogonal factor (cheaper than
// back-transforming n vectors one reflector at a time), then the recorded rotations applied to it by cache-resident
// row blocks instead of column pair by column pair.  U: n x n, D: n, both ASCENDING like sym_eigen.
bool sym_eigen_blocked(const double* M, int n, double* U, double* D) {
    if (n <= 2) return sym_eigen(M, n, U, D);
    std::vector<double> V((size_t)n * n), d(n), e(n);
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) V[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : M[(size_t)r * n + c];
    tridiagonalize(n, V.data(), d.data(), e.data());
    std::vector<Sweep> sweeps;
    std::vector<double> cs, sn;
    cs.reserve((size_t)n * n);
    sn.reserve((size_t)n * n);
    if (!ql_record(n, d.data(), e.data(), sweeps, cs, sn)) return false;
    const int ldz = (n + 7) & ~7;
    std::vector<double> Zbuf((size_t)ldz * n + 8, 0.0);
    double* Z = Zbuf.data();
    while (reinterpret_cast<uintptr_t>(Z) & 63) ++Z;
    for (int j = 0; j < n; ++j) std::copy(V.begin() + (size_t)j * n, V.begin() + (size_t)j * n + n, Z + (size_t)j * ldz);
    const int nvec = ldz / 8;
    for (int b = 0; b * 8 < nvec; ++b)
        apply_rotations_rows(ldz, Z, b * 64, std::min(8, nvec - b * 8), sweeps.data(), sweeps.size(), cs.data(), sn.data());
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
    for (int j = 0; j < n; ++j) {
        D[j] = d[idx[j]];
        std::copy(Z + (size_t)idx[j] * ldz, Z + (size_t)idx[j] * ldz + n, U + (size_t)j * n);
    }
    return true;
}

// eigenDecomposition (src/filter.cpp:204-228) when only the leading kmax eigenpairs and the count of the cut are used
// (orthogonalize on Q, :313-316): Dk[0 .. kmax) = the kmax largest eigenvalues DESCENDING, U (n x kmax) their eigenvectors,
// *r_out = number of eigenvalues >= eps.  Householder reduction, eigenvalues by bisection, eigenvectors by inverse iteration
// and back-transformation.  For 2 kmax <= n (else, and should the inverse iteration give up: eigen_decomposition_top).
bool eigen_decomposition_topk(const double* M, int n, double eps, int kmax, double* U, double* Dk, int* r_out) {
    kmax = std::max(0, std::min(kmax, n));
    if (n < 8 || 2 * kmax > n || std::getenv("NLE_EIG_NO_BISECT") != nullptr) {
        std::vector<double> D(n);
        if (!eigen_decomposition_top(M, n, eps, kmax, U, D.data(), r_out)) return false;
        std::copy(D.begin(), D.begin() + kmax, Dk);
        return true;
    }
    static const bool trace = std::getenv("NLE_EIG_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0.0;
    std::vector<double> V((size_t)n * n), d(n), e(n), hs(n);
    for (int c = 0; c < n; ++c)  // mirror the lower triangle (SelfAdjointEigenSolver reads only the lower one)
        for (int r = 0; r < n; ++r) V[(size_t)c * n + r] = (r >= c) ? M[(size_t)c * n + r] : M[(size_t)r * n + c];
    tridiag_reduce(n, V.data(), d.data(), e.data(), hs.data());
    const double t1 = trace ? now() : 0.0;
    const SturmT t = sturm_setup(n, d.data(), e.data());
    *r_out = n - sturm_count1(t, eps);  // descending order: the leading run >= eps IS the count of eigenvalues >= eps
    sturm_eigenvalues_desc(t, 0, kmax, Dk);
    const double t2 = trace ? now() : 0.0;
    if (kmax == 0) return true;
    if (!tridiag_inverse_iteration(n, d.data(), e.data(), Dk, kmax, U)) {
        std::vector<double> D(n);
        if (!eigen_decomposition_top(M, n, eps, kmax, U, D.data(), r_out)) return false;
        std::copy(D.begin(), D.begin() + kmax, Dk);
        return true;
    }
    const double t3 = trace ? now() : 0.0;
    back_transform_cols(n, V.data(), hs.data(), U, 0, kmax);
    if (trace)
        std::fprintf(stderr, "[nle eig] n = %d, %d pairs: reduction %.3f ms, bisection %.3f ms, inverse iteration %.3f ms, back-transformation %.3f ms\n",
                     n, kmax, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    return true;
}

bool eigen_decomposition_top(const double* M, int n, double eps, int kmax, double* U, double* D, int* r_out) {
    kmax = std::max(0, std::min(kmax, n));
    if (2 * kmax > n && default_threads(n, kmax) == 1) {
        // most eigenvectors wanted, one thread: accumulating the orthogonal factor (classic form) is cheaper
        // than back-transforming them one by one (n = 200: 1.65 ms against 1.9 ms; n = 900: 116 against 124 ms,
        // but ~70 ms once the two phases are threaded)
        std::vector<double> Ua((size_t)n * n), Da(n);
        if (!sym_eigen_blocked(M, n, Ua.data(), Da.data())) return false;
        for (int j = 0; j < n; ++j) {  // ascending -> descending
            const int src = n - 1 - j;
            D[j] = Da[src];
            if (j < kmax) std::copy(Ua.begin() + (size_t)src * n, Ua.begin() + (size_t)src * n + n, U + (size_t)j * n);
        }
    } else if (!sym_eigen_top(M, n, kmax, 0, U, D)) {
        return false;
    }
    // reference src/filter.cpp:209-216: descending, keep the leading run >= eps
    int r = 0;
    while (r < n && D[r] >= eps) ++r;
    *r_out = r;
    return true;
}

bool eigen_decomposition(const double* M, int n, double eps, double* U, double* D, int* r_out) {
    return eigen_decomposition_top(M, n, eps, n, U, D, r_out);
}

// ---- opt-in top-K solver with the semantics of the reference's USE_SPECTRA build (src/filter.cpp:170-199) ----
// Spectra::SymEigsSolver<double, LARGEST_MAGN, DenseGenMatProd> (ext/Spectra/SymEigsBase.h): nev = min(K, n - 1),
// ncv = min(2 nev, n), tolerance 1e-10 on the Ritz residual |beta s_mi| < tol max(eps^(2/3), |theta_i|), at most 1000
// restarts, the wanted set is the nev Ritz values of LARGEST MAGNITUDE, the FULL matrix is multiplied (no triangle is
// mirrored: Q is only symmetric up to the asymmetry of Wa), only converged pairs are returned, sorted by algebraic
// value descending; the caller then keeps the leading run >= eps (:186-196).
// Restarting is thick (Wu & Simon) instead of Spectra's implicit QR shifts: the two span the same Krylov subspaces in
// exact arithmetic, and the converged eigenpairs agree to the tolerance either way.  Full reorthogonalisation, twice.
int lanczos_topk(const double* A, int n, int nev_in, double tol, int max_restarts, double* U, double* D, int* restarts_out) {
    if (n <= 1 || nev_in < 1) return 0;
    const int nev = std::min(nev_in, n - 1), m = std::min(2 * nev, n);
    const double eps23 = std::pow(2.220446049250313e-16, 2.0 / 3.0);
    std::vector<double> V((size_t)n * (m + 1), 0.0), T((size_t)m * m, 0.0), w(n), S((size_t)m * m), theta(m), Vn((size_t)n * m);
    // deterministic start vector (Spectra seeds its own generator with 0; any start reaches the same eigenpairs)
    {
        unsigned long long x = 0x9E3779B97F4A7C15ull;
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            V[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            nrm += V[i] * V[i];
        }
        nrm = std::sqrt(nrm);
        for (int i = 0; i < n; ++i) V[i] /= nrm;
    }
    auto col = [&](int j) { return V.data() + (size_t)j * n; };
    int k = 0, restarts = 0, nconv = 0;
    double beta_m = 0.0;
    std::vector<int> order(m);
    for (;;) {
        // extend the factorisation A V_j = V_j T_j + beta_j v_j e_j^T from k to m columns
        for (int j = k; j < m; ++j) {
            const double* vj = col(j);
            for (int i = 0; i < n; ++i) w[i] = 0.0;
            for (int c = 0; c < n; ++c) {  // w = A v_j (column-major full matrix)
                const double x = vj[c];
                const double* a = A + (size_t)c * n;
#pragma omp simd
                for (int i = 0; i < n; ++i) w[i] += a[i] * x;
            }
            for (int pass = 0; pass < 2; ++pass)  // Gram-Schmidt against every basis vector, twice
                for (int c = 0; c <= j; ++c) {
                    const double* vc = col(c);
                    double h = 0.0;
#pragma omp simd reduction(+ : h)
                    for (int i = 0; i < n; ++i) h += vc[i] * w[i];
                    if (pass == 0 && (c == j || (j == k && c < k) || c == j - 1)) {
                        if (c == j) T[(size_t)j * m + j] += h;  // alpha_j
                    }
#pragma omp simd
                    for (int i = 0; i < n; ++i) w[i] -= h * vc[i];
                }
            double b = 0.0;
            for (int i = 0; i < n; ++i) b += w[i] * w[i];
            b = std::sqrt(b);
            if (j + 1 < m) {
                T[(size_t)j * m + (j + 1)] = b;
                T[(size_t)(j + 1) * m + j] = b;
            } else {
                beta_m = b;
            }
            double* vn = col(j + 1);
            if (b > 0.0) {
                for (int i = 0; i < n; ++i) vn[i] = w[i] / b;
            } else {  // invariant subspace: any unit vector orthogonal to the basis continues the recursion
                for (int i = 0; i < n; ++i) vn[i] = 0.0;
            }
        }
        // Ritz pairs of the projected matrix (dense symmetric after a thick restart: diagonal + arrow + tridiagonal)
        if (!sym_eigen(T.data(), m, S.data(), theta.data())) return -1;
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return std::fabs(theta[a]) > std::fabs(theta[b2]); });
        nconv = 0;
        for (int i = 0; i < nev; ++i) {
            const int c = order[i];
            const double resid = std::fabs(beta_m * S[(size_t)c * m + (m - 1)]);
            if (resid < tol * std::max(eps23, std::fabs(theta[c]))) ++nconv;
        }
        if (nconv >= nev || restarts >= max_restarts || m >= n) break;
        ++restarts;
        // thick restart: keep nev + min(nconv, (m - nev) / 2) Ritz vectors (Spectra's nev_adjusted), at most m - 1
        int keep = nev + std::min(nconv, (m - nev) / 2);
        keep = std::max(1, std::min(keep, m - 1));
        for (int c = 0; c < keep; ++c) {
            const double* sc = S.data() + (size_t)order[c] * m;
            double* out = Vn.data() + (size_t)c * n;
            for (int i = 0; i < n; ++i) out[i] = 0.0;
            for (int j = 0; j < m; ++j) {
                const double x = sc[j];
                const double* vj = col(j);
#pragma omp simd
                for (int i = 0; i < n; ++i) out[i] += vj[i] * x;
            }
        }
        std::vector<double> tk(keep), sk(keep);
        for (int c = 0; c < keep; ++c) {
            tk[c] = theta[order[c]];
            sk[c] = beta_m * S[(size_t)order[c] * m + (m - 1)];
        }
        std::copy(Vn.begin(), Vn.begin() + (size_t)keep * n, V.begin());
        std::copy(col(m), col(m) + n, col(keep));  // the residual direction becomes basis vector `keep`
        std::fill(T.begin(), T.end(), 0.0);
        for (int c = 0; c < keep; ++c) {
            T[(size_t)c * m + c] = tk[c];
            T[(size_t)keep * m + c] = sk[c];
            T[(size_t)c * m + keep] = sk[c];
        }
        k = keep;
    }
    if (restarts_out) *restarts_out = restarts;
    // converged wanted pairs, algebraic value descending
    std::vector<int> sel;
    for (int i = 0; i < nev; ++i) {
        const int c = order[i];
        const double resid = std::fabs(beta_m * S[(size_t)c * m + (m - 1)]);
        if (m >= n || resid < tol * std::max(eps23, std::fabs(theta[c]))) sel.push_back(c);
    }
    std::stable_sort(sel.begin(), sel.end(), [&](int a, int b2) { return theta[a] > theta[b2]; });
    for (size_t q = 0; q < sel.size(); ++q) {
        D[q] = theta[sel[q]];
        const double* sc = S.data() + (size_t)sel[q] * m;
        double* out = U + q * (size_t)n;
        for (int i = 0; i < n; ++i) out[i] = 0.0;
        for (int j = 0; j < m; ++j) {
            const double x = sc[j];
            const double* vj = col(j);
#pragma omp simd
            for (int i = 0; i < n; ++i) out[i] += vj[i] * x;
        }
    }
    return (int)sel.size();
}

}  // namespace nleh
