// CPU BASELINE / ORACLE IN C++ -- TEST AND MEASUREMENT INFRASTRUCTURE ONLY (same rules as oracle/nle_oracle.py: only tests/,
// __graft_entry__ and bench.py's cpu_baseline leg may build or run this; the product never does).
//
// A plain C++17 + OpenMP fp64 restatement of the hot path of lightalchemist/nonlocal-image-edit in the STREAMING form of
// SURVEY.md section 7 step 2 (the same decomposition as oracle/nle_oracle.py: train_filter_streaming, which is pinned by the
// reference's own test cases and README pairs; this file is pinned against that oracle by tests/test_cpu_baseline.py):
// natural pixel order, pixel tiles over OpenMP threads, Phi (N x r fp64) held in memory like the reference holds its
// N x p matrices.  Follows src/filter.cpp stage by stage:
//   :56-80    samplePixels (closed form)            :104-145  affinities exp(-d^2/hx^2 - dv^2/hy^2), integer spatial term
//   :204-228  eigenDecomposition (lower triangle, descending, cut at 1e-10) -- Householder + implicit QL, written here
//   :257-280  Phi = [V_A ; K_AB^T V_A Lambda^-1]     :230-254  Sinkhorn, W blocks with q = phi.cols()
//   :282-331  orthogonalize                          :334-347, :445-458  layer responses, apply
// It exists to put a real "reference-class CPU path" number beside the GPU's in bench.py: the reference itself cannot be
// built here (Eigen3 and OpenCV are absent: SURVEY.md section 8c) and is single threaded (CMakeLists.txt:40-46), so the
// baseline is reported for 1 thread and for all cores.
//
//   nle_cpu_baseline H W nRow nCol hx hy T K L threads [out.json]
// Synthetic luminance of SURVEY.md section 8d (seed 1234).  Prints one JSON line: seconds per stage, eigenvalues, layer norms,
// a few probe values.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <omp.h>

namespace {
constexpr double kEps = 1e-10;  // include/filter.hpp:14
using Vec = std::vector<double>;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- synthetic input (bit-identical to oracle/nle_oracle.py: synthetic_luminance)
uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void synthetic(int H, int W, Vec& x) {
    x.resize((size_t)H * W);
    const double two_pi = 2.0 * M_PI;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const double a = (double)r / H, b = (double)c / W;
            const double s = 0.5 * std::sin(two_pi * (1.5 * a + 0.5 * b)) + 0.5 * std::cos(two_pi * (0.7 * a - 2.2 * b));
            const uint64_t h = splitmix64(((uint64_t)r * (uint64_t)W + (uint64_t)c) ^ 1234ull);
            const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
            double v = std::nearbyint(128.0 + 70.0 * s + 40.0 * (u - 0.5));
            x[(size_t)r * W + c] = std::min(255.0, std::max(0.0, v));
        }
}

// ---- symmetric eigensolver: Householder tridiagonalisation with accumulated transformations + implicit QL (the classic
// tred2 / tql2 scheme).  A: n x n row-major symmetric (LOWER triangle read, :207).  Out: eigenvalues DESCENDING, eigenvectors
// as columns of U (row-major n x n), :209-212.
bool sym_eig_desc(const Vec& A, int n, Vec& U, Vec& D) {
    Vec V((size_t)n * n), d(n), e(n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i >= j) ? A[(size_t)i * n + j] : A[(size_t)j * n + i];
    auto v = [&](int i, int j) -> double& { return V[(size_t)i * n + j]; };
    for (int j = 0; j < n; ++j) d[j] = v(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) {
                d[j] = v(i - 1, j);
                v(i, j) = 0.0;
                v(j, i) = 0.0;
            }
        } else {
            for (int k = 0; k < i; ++k) {
                d[k] /= scale;
                h += d[k] * d[k];
            }
            double f = d[i - 1], g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                v(j, i) = f;
                g = e[j] + v(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) {
                    g += v(k, j) * d[k];
                    e[k] += v(k, j) * f;
                }
                e[j] = g;
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
                for (int k = j; k <= i - 1; ++k) v(k, j) -= (f * e[k] + g * d[k]);
                d[j] = v(i - 1, j);
                v(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        v(n - 1, i) = v(i, i);
        v(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = v(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += v(k, i + 1) * v(k, j);
                for (int k = 0; k <= i; ++k) v(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) v(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) {
        d[j] = v(n - 1, j);
        v(n - 1, j) = 0.0;
    }
    v(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    // implicit QL
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
                double r = std::hypot(p, 1.0);
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
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = v(k, i + 1);
                        v(k, i + 1) = s * v(k, i) + c * h;
                        v(k, i) = c * v(k, i) - s * h;
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
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] > d[b]; });
    U.assign((size_t)n * n, 0.0);
    D.resize(n);
    for (int j = 0; j < n; ++j) {
        D[j] = d[idx[j]];
        for (int i = 0; i < n; ++i) U[(size_t)i * n + j] = v(i, idx[j]);
    }
    return true;
}
// eigenDecomposition (:204-228): leading run >= eps kept; returns r
int eig_cut(const Vec& A, int n, Vec& U, Vec& D) {
    if (!sym_eig_desc(A, n, U, D)) {
        std::fprintf(stderr, "eigensolver did not converge\n");
        std::exit(2);
    }
    int r = 0;
    while (r < n && D[r] >= kEps) ++r;  // :214
    return r;
}
inline double recip0(double v) { return std::fabs(v) >= kEps ? 1.0 / v : 0.0; }  // :42-54

// C (m x n) (+)= A (m x k) B (k x n), all row-major.  Register-blocked 4 x 16 micro-kernel on 256-bit vectors (GCC vector
// extensions: 16 accumulators, 4 loads of B and 4 broadcasts of A per 16 multiply-adds) -- what a BLAS-backed Eigen product
// comes to on one core, within a small factor; tails in 4-wide and scalar code.
typedef double v4d __attribute__((vector_size(32)));
typedef double v4du __attribute__((vector_size(32), aligned(8)));
void mm(const double* A, const double* B, double* C, int m, int k, int n, bool accumulate = false) {
    const int n16 = n & ~15, n4 = n & ~3;
    for (int i0 = 0; i0 < m; i0 += 4) {
        const int mi = std::min(4, m - i0);
        for (int j0 = 0; j0 < n16; j0 += 16) {
            v4d acc[4][4];
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
            for (int q = 0; q < k; ++q) {
                const double* bq = B + (size_t)q * n + j0;
                const v4d b0 = *(const v4du*)(bq), b1 = *(const v4du*)(bq + 4), b2 = *(const v4du*)(bq + 8), b3 = *(const v4du*)(bq + 12);
                for (int a = 0; a < 4; ++a) {
                    const double av = a < mi ? A[(size_t)(i0 + a) * k + q] : 0.0;
                    const v4d va = v4d{av, av, av, av};
                    acc[a][0] += va * b0;
                    acc[a][1] += va * b1;
                    acc[a][2] += va * b2;
                    acc[a][3] += va * b3;
                }
            }
            for (int a = 0; a < mi; ++a)
                for (int b = 0; b < 4; ++b) {
                    v4du* c = (v4du*)(C + (size_t)(i0 + a) * n + j0 + 4 * b);
                    *c = accumulate ? (v4d)(*c + acc[a][b]) : acc[a][b];
                }
        }
        for (int j0 = n16; j0 < n4; j0 += 4) {
            v4d acc[4];
            for (int a = 0; a < 4; ++a) acc[a] = v4d{0.0, 0.0, 0.0, 0.0};
            for (int q = 0; q < k; ++q) {
                const v4d b0 = *(const v4du*)(B + (size_t)q * n + j0);
                for (int a = 0; a < mi; ++a) {
                    const double av = A[(size_t)(i0 + a) * k + q];
                    acc[a] += v4d{av, av, av, av} * b0;
                }
            }
            for (int a = 0; a < mi; ++a) {
                v4du* c = (v4du*)(C + (size_t)(i0 + a) * n + j0);
                *c = accumulate ? (v4d)(*c + acc[a]) : acc[a];
            }
        }
        for (int j = n4; j < n; ++j)
            for (int a = 0; a < mi; ++a) {
                double sacc = 0.0;
                for (int q = 0; q < k; ++q) sacc += A[(size_t)(i0 + a) * k + q] * B[(size_t)q * n + j];
                C[(size_t)(i0 + a) * n + j] = accumulate ? C[(size_t)(i0 + a) * n + j] + sacc : sacc;
            }
    }
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 11) {
        std::fprintf(stderr, "usage: nle_cpu_baseline H W nRow nCol hx hy T K L threads [out.json]\n");
        return 1;
    }
    const int H = std::atoi(argv[1]), W = std::atoi(argv[2]), nRow = std::atoi(argv[3]), nCol = std::atoi(argv[4]);
    const double hx = std::atof(argv[5]), hy = std::atof(argv[6]);
    const int T = std::atoi(argv[7]), Kreq = std::atoi(argv[8]), L = std::atoi(argv[9]), threads = std::max(1, std::atoi(argv[10]));
    omp_set_num_threads(threads);
    const long long N = (long long)H * W;
    Vec x;
    synthetic(H, W, x);
    const double t_begin = now_s();

    // samplePixels (:56-80), closed form
    const int rowStep = H / nRow, colStep = W / nCol;
    const int rowOff = (rowStep - 1 + (H - rowStep * nRow)) / 2, colOff = (colStep - 1 + (W - colStep * nCol)) / 2;
    std::vector<int> selR, selC;
    for (int r = 0; r < H; ++r)
        if (r >= rowOff && r <= H - rowOff && (r - rowOff) % rowStep == 0) selR.push_back(r);
    for (int c = 0; c < W; ++c)
        if (c >= colOff && c <= W - colOff && (c - colOff) % colStep == 0) selC.push_back(c);
    const int p = (int)(selR.size() * selC.size());
    std::vector<int> sr(p), sc(p);
    Vec sv(p);
    std::vector<long long> spix(p);
    for (size_t a = 0; a < selR.size(); ++a)
        for (size_t b = 0; b < selC.size(); ++b) {
            const int k = (int)(a * selC.size() + b);
            sr[k] = selR[a];
            sc[k] = selC[b];
            spix[k] = (long long)selR[a] * W + selC[b];
            sv[k] = x[spix[k]];
        }
    std::vector<char> is_sample((size_t)N, 0);
    for (int k = 0; k < p; ++k) is_sample[spix[k]] = 1;
    const double sw = 1.0 / (hx * hx), pw = 1.0 / (hy * hy);  // :128-129

    // Ka and its eigenpairs (:133-137, :262-271)
    Vec Ka((size_t)p * p);
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j) {
            const long long dr = sr[i] - sr[j], dc = sc[i] - sc[j];
            const double dv = sv[i] - sv[j];
            Ka[(size_t)i * p + j] = std::exp(-sw * (double)(dr * dr + dc * dc) - pw * dv * dv);
        }
    Vec VA, lam;
    const int r = eig_cut(Ka, p, VA, lam);  // VA: p x p row-major, first r columns kept
    Vec B((size_t)p * r);                   // V_A Lambda^-1 (:268, :275)
    for (int a = 0; a < p; ++a)
        for (int k = 0; k < r; ++k) B[(size_t)a * r + k] = VA[(size_t)a * p + k] * recip0(lam[k]);
    const double t_ka = now_s();

    // Phi (N x r), natural pixel order; sample rows = exact V_A rows (:275 top block)
    Vec phi((size_t)N * r);
    const int TILE = 256;
#pragma omp parallel
    {
        Vec kt((size_t)TILE * p);
#pragma omp for schedule(dynamic, 4)
        for (long long t0 = 0; t0 < N; t0 += TILE) {
            const int m = (int)std::min<long long>(TILE, N - t0);
            for (int i = 0; i < m; ++i) {
                const long long gi = t0 + i;
                const int row = (int)(gi / W), col = (int)(gi - (long long)row * W);
                const double xv = x[gi];
                double* kr = kt.data() + (size_t)i * p;
                for (int s = 0; s < p; ++s) {
                    const long long dr = row - sr[s], dc = col - sc[s];
                    const double dv = xv - sv[s];
                    kr[s] = std::exp(-sw * (double)(dr * dr + dc * dc) - pw * dv * dv);  // :104-112, :145
                }
            }
            mm(kt.data(), B.data(), phi.data() + (size_t)t0 * r, m, p, r);
        }
    }
    for (int k = 0; k < p; ++k)
        for (int j = 0; j < r; ++j) phi[(size_t)spix[k] * r + j] = VA[(size_t)k * p + j];
    const double t_phi = now_s();

    // Sinkhorn (:238-245): one fused pass per half-iteration: y_i = recip(phi_i . u), t += phi_i y_i
    auto pass = [&](const Vec* u, Vec& y, Vec& tsum) {  // u == nullptr: y = 1 (the initial r)
        tsum.assign(r, 0.0);
#pragma omp parallel
        {
            Vec tl(r, 0.0);
#pragma omp for schedule(static)
            for (long long i = 0; i < N; ++i) {
                const double* ph = phi.data() + (size_t)i * r;
                double yi = 1.0;
                if (u) {
                    const double* uu = u->data();
                    v4d a0 = v4d{0.0, 0.0, 0.0, 0.0}, a1 = a0;
                    int j = 0;
                    for (; j + 8 <= r; j += 8) {
                        a0 += *(const v4du*)(ph + j) * *(const v4du*)(uu + j);
                        a1 += *(const v4du*)(ph + j + 4) * *(const v4du*)(uu + j + 4);
                    }
                    a0 += a1;
                    double s = (a0[0] + a0[1]) + (a0[2] + a0[3]);
                    for (; j < r; ++j) s += ph[j] * uu[j];
                    yi = recip0(s);
                }
                y[i] = yi;
                {
                    const v4d vy = v4d{yi, yi, yi, yi};
                    double* tt = tl.data();
                    int j = 0;
                    for (; j + 4 <= r; j += 4) *(v4du*)(tt + j) = (v4d)(*(v4du*)(tt + j) + *(const v4du*)(ph + j) * vy);
                    for (; j < r; ++j) tt[j] += ph[j] * yi;
                }
            }
#pragma omp critical
            for (int j = 0; j < r; ++j) tsum[j] += tl[j];
        }
    };
    Vec y((size_t)N), cvec((size_t)N), tsum, u_c(r), u_r(r);
    pass(nullptr, y, tsum);
    for (int it = 0; it < T; ++it) {
        for (int j = 0; j < r; ++j) u_c[j] = lam[j] * tsum[j];
        pass(&u_c, cvec, tsum);  // c
        for (int j = 0; j < r; ++j) u_r[j] = lam[j] * tsum[j];
        pass(&u_r, y, tsum);     // r
    }
    const double t_sink = now_s();

    // W blocks (:247-250) with q = r: A block = the first q samples
    const int q = r;
    Vec cA(q), rA(q), left((size_t)q * r), Wa((size_t)q * q);
    for (int a = 0; a < q; ++a) {
        double sc_ = 0.0, sr_ = 0.0;
        for (int j = 0; j < r; ++j) {
            sc_ += VA[(size_t)a * p + j] * u_c[j];
            sr_ += VA[(size_t)a * p + j] * u_r[j];
        }
        cA[a] = recip0(sc_);
        rA[a] = recip0(sr_);
        for (int j = 0; j < r; ++j) left[(size_t)a * r + j] = rA[a] * VA[(size_t)a * p + j] * lam[j];
    }
    for (int a = 0; a < q; ++a)
        for (int b = 0; b < q; ++b) {
            double s = 0.0;
            for (int j = 0; j < r; ++j) s += left[(size_t)a * r + j] * cA[b] * VA[(size_t)b * p + j];
            Wa[(size_t)a * q + b] = s;  // :249
        }
    // Gram over the B block: G = sum_{i not in A} c_i^2 phi_i phi_i^T (:296, Wab Wab^T = M^T G M)
    std::vector<char> inA((size_t)N, 0);
    for (int a = 0; a < q; ++a) inA[spix[a]] = 1;
    Vec G((size_t)r * r, 0.0);
#pragma omp parallel
    {
        Vec Gl((size_t)r * r, 0.0), Z((size_t)TILE * r), Zt((size_t)r * TILE);
#pragma omp for schedule(static)
        for (long long t0 = 0; t0 < N; t0 += TILE) {
            const int m = (int)std::min<long long>(TILE, N - t0);
            for (int i = 0; i < m; ++i) {
                const double ci = inA[t0 + i] ? 0.0 : cvec[t0 + i];
                const double* ph = phi.data() + (size_t)(t0 + i) * r;
                for (int j = 0; j < r; ++j) {
                    const double zv = ci * ph[j];
                    Z[(size_t)i * r + j] = zv;
                    Zt[(size_t)j * m + i] = zv;
                }
            }
            mm(Zt.data(), Z.data(), Gl.data(), r, m, r, /*accumulate=*/true);  // G += Z^T Z
        }
#pragma omp critical
        for (size_t t = 0; t < G.size(); ++t) G[t] += Gl[t];
    }
    const double t_gram = now_s();

    // orthogonalize (:282-331)
    Vec U2, l2;
    const int r2 = eig_cut(Wa, q, U2, l2);
    Vec S((size_t)q * q, 0.0);
    for (int a = 0; a < q; ++a)
        for (int b = 0; b < q; ++b) {
            double s = 0.0;
            for (int k = 0; k < r2; ++k) s += U2[(size_t)a * q + k] * std::sqrt(recip0(l2[k])) * U2[(size_t)b * q + k];
            S[(size_t)a * q + b] = s;  // :289-292
        }
    // M = diag(lam) Phi_A^T diag(rA) = left^T  (r x q);  WW = M^T G M (q x q)
    Vec GM((size_t)r * q), WW((size_t)q * q), T1((size_t)q * q), Q((size_t)q * q);
    {
        Vec M((size_t)r * q);
        for (int j = 0; j < r; ++j)
            for (int a = 0; a < q; ++a) M[(size_t)j * q + a] = left[(size_t)a * r + j];
        mm(G.data(), M.data(), GM.data(), r, r, q);
        mm(left.data(), GM.data(), WW.data(), q, r, q);
    }
    mm(S.data(), WW.data(), T1.data(), q, q, q);
    mm(T1.data(), S.data(), Q.data(), q, q, q);
    for (size_t t = 0; t < Q.size(); ++t) Q[t] += Wa[t];  // :296
    Vec Vq, Sq;
    const int rq = eig_cut(Q, q, Vq, Sq);
    const int K = std::min(Kreq, rq);  // :314
    Vec T2((size_t)q * K);             // S Vq Sq^-1/2
    for (int a = 0; a < q; ++a)
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int b = 0; b < q; ++b) s += S[(size_t)a * q + b] * Vq[(size_t)b * q + k];
            T2[(size_t)a * K + k] = s * std::sqrt(recip0(Sq[k]));
        }
    Vec Cproj((size_t)r * K), WaT2((size_t)q * K);
    {
        Vec M((size_t)r * q);
        for (int j = 0; j < r; ++j)
            for (int a = 0; a < q; ++a) M[(size_t)j * q + a] = left[(size_t)a * r + j];
        mm(M.data(), T2.data(), Cproj.data(), r, q, K);
        mm(Wa.data(), T2.data(), WaT2.data(), q, q, K);
    }
    const double t_ortho = now_s();

    // V (N x K) = diag(c) Phi Cproj; A rows = Wa T2 (:324-327)
    Vec V((size_t)N * K);
#pragma omp parallel for schedule(static)
    for (long long t0 = 0; t0 < N; t0 += TILE) {
        const int m = (int)std::min<long long>(TILE, N - t0);
        mm(phi.data() + (size_t)t0 * r, Cproj.data(), V.data() + (size_t)t0 * K, m, r, K);
        for (int i = 0; i < m; ++i) {
            const double ci = inA[t0 + i] ? 0.0 : cvec[t0 + i];
            for (int k = 0; k < K; ++k) V[(size_t)(t0 + i) * K + k] *= ci;
        }
    }
    for (int a = 0; a < q; ++a)
        for (int k = 0; k < K; ++k) V[(size_t)spix[a] * K + k] = WaT2[(size_t)a * K + k];
    const double t_proj = now_s();

    // apply (:445-458) per layer (:334-347)
    Vec tv(K, 0.0);
#pragma omp parallel
    {
        Vec tl(K, 0.0);
#pragma omp for schedule(static)
        for (long long i = 0; i < N; ++i)
            for (int k = 0; k < K; ++k) tl[k] += V[(size_t)i * K + k] * x[i];
#pragma omp critical
        for (int k = 0; k < K; ++k) tv[k] += tl[k];
    }
    Vec norms(L, 0.0), probes((size_t)L * 8, 0.0);
    std::vector<float> layer((size_t)N);
    for (int l = 0; l < L; ++l) {
        Vec g(K);
        for (int k = 0; k < K; ++k) {
            const double lk = Sq[k];
            g[k] = ((l < L - 1) ? (std::pow(lk, (double)l) - std::pow(lk, (double)(l + 1))) : std::pow(lk, (double)(L - 1))) * tv[k];
        }
        double n2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : n2)
        for (long long i = 0; i < N; ++i) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += V[(size_t)i * K + k] * g[k];
            layer[i] = (float)s;
            n2 += s * s;
        }
        norms[l] = std::sqrt(n2);
        for (int j = 0; j < 8; ++j) probes[(size_t)l * 8 + j] = layer[(size_t)((N - 1) * j / 7)];
    }
    const double t_end = now_s();

    std::string js = "{";
    char buf[256];
    std::snprintf(buf, sizeof buf, "\"H\": %d, \"W\": %d, \"p\": %d, \"r\": %d, \"r_wa\": %d, \"r_q\": %d, \"K\": %d, \"threads\": %d, ", H, W, p, r,
                  r2, rq, K, threads);
    js += buf;
    std::snprintf(buf, sizeof buf,
                  "\"seconds\": %.4f, \"stages\": {\"ka\": %.4f, \"phi\": %.4f, \"sinkhorn\": %.4f, \"gram\": %.4f, \"ortho\": %.4f, "
                  "\"project\": %.4f, \"apply\": %.4f}, ",
                  t_end - t_begin, t_ka - t_begin, t_phi - t_ka, t_sink - t_phi, t_gram - t_sink, t_ortho - t_gram, t_proj - t_ortho,
                  t_end - t_proj);
    js += buf;
    js += "\"eigvals\": [";
    for (int k = 0; k < K; ++k) {
        std::snprintf(buf, sizeof buf, "%s%.17g", k ? ", " : "", Sq[k]);
        js += buf;
    }
    js += "], \"layer_norms\": [";
    for (int l = 0; l < L; ++l) {
        std::snprintf(buf, sizeof buf, "%s%.17g", l ? ", " : "", norms[l]);
        js += buf;
    }
    js += "], \"probes\": [";
    for (size_t t = 0; t < probes.size(); ++t) {
        std::snprintf(buf, sizeof buf, "%s%.9g", t ? ", " : "", probes[t]);
        js += buf;
    }
    js += "]}";
    std::puts(js.c_str());
    if (argc > 11) {
        FILE* fh = std::fopen(argv[11], "w");
        if (fh) {
            std::fputs(js.c_str(), fh);
            std::fclose(fh);
        }
    }
    return 0;
}
