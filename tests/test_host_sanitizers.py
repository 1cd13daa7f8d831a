"""ASan + UBSan build of the host-only fp64 algebra (csrc/eigen_sym.cpp) with a small driver: the eigensolver
is the one piece of native host code with hand-written index arithmetic.  (GPU AddressSanitizer is not
available on this pool, so sanitizers run on the CPU build only.)"""
import os
import shutil
import subprocess
import textwrap

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "nonlocal-image-edit_amd", "csrc")

DRIVER = textwrap.dedent(r"""
    #include <cmath>
    #include <cstdio>
    #include <random>
    #include <vector>
    #include "eigen_sym.h"
    int main() {
        std::mt19937 g(7);
        std::normal_distribution<double> N(0, 1);
        int bad = 0;
        for (int n : {1, 2, 3, 5, 17, 64, 130}) {
            std::vector<double> B((size_t)n * n), A((size_t)n * n, 0.0), U((size_t)n * n), D(n);
            for (auto& v : B) v = N(g);
            const int rank = n > 4 ? n - 2 : n;  // rank deficient for larger n: exercises the 1e-10 cut
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0;
                    for (int k = 0; k < rank; ++k) s += B[i + (size_t)k * n] * B[j + (size_t)k * n];
                    A[i + (size_t)j * n] = s / n;
                }
            int r = 0;
            if (!nleh::eigen_decomposition(A.data(), n, 1e-10, U.data(), D.data(), &r)) { ++bad; continue; }
            if (r != rank) { std::printf("n=%d: rank %d expected %d\n", n, r, rank); ++bad; }
            for (int k = 1; k < r; ++k) if (D[k] > D[k - 1] + 1e-12) ++bad;          // descending
            double err = 0;                                                           // U D U^T == A on the kept part
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0;
                    for (int k = 0; k < r; ++k) s += U[i + (size_t)k * n] * D[k] * U[j + (size_t)k * n];
                    err = std::fmax(err, std::fabs(s - A[i + (size_t)j * n]));
                }
            if (err > 1e-9) { std::printf("n=%d: reconstruction error %g\n", n, err); ++bad; }
        }
        // Cholesky + inverse + the trace certificate, and the small products
        for (int n : {1, 2, 7, 33, 100}) {
            std::vector<double> B((size_t)n * n), A((size_t)n * n, 0.0), L((size_t)n * n), Li((size_t)n * n), P((size_t)n * n);
            for (auto& v : B) v = N(g);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = (i == j) ? 0.5 : 0.0;
                    for (int k = 0; k < n; ++k) s += B[i + (size_t)k * n] * B[j + (size_t)k * n];
                    A[i + (size_t)j * n] = s / n;
                }
            double tr = 0;
            if (!nleh::cholesky_with_inverse(A.data(), n, L.data(), Li.data(), &tr)) { std::printf("n=%d: cholesky failed\n", n); ++bad; continue; }
            nleh::gemm_nt_cols(L.data(), L.data(), P.data(), n, n, n, 0, n);   // L L^T
            double err = 0;
            for (size_t i = 0; i < P.size(); ++i) err = std::fmax(err, std::fabs(P[i] - A[i]));
            nleh::gemm_nn_cols(Li.data(), L.data(), P.data(), n, n, n, 0, n);  // L^-1 L
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) err = std::fmax(err, std::fabs(P[i + (size_t)j * n] - (i == j)));
            nleh::gemm_tn_cols(Li.data(), Li.data(), P.data(), n, n, n, 0, n);  // A^-1 = L^-T L^-1
            double t2 = 0;
            for (int i = 0; i < n; ++i) t2 += P[i + (size_t)i * n];
            if (err > 1e-9 || std::fabs(t2 - tr) > 1e-9 * tr) { std::printf("n=%d: cholesky error %g trace %g vs %g\n", n, err, tr, t2); ++bad; }
            std::vector<double> U((size_t)n * n), D(n);
            int r = 0;
            nleh::eigen_decomposition(A.data(), n, 0.0, U.data(), D.data(), &r);
            if (D[n - 1] < 1.0 / tr * (1 - 1e-9)) { std::printf("n=%d: lambda_min %g below the certificate %g\n", n, D[n - 1], 1.0 / tr); ++bad; }
        }
        // the three-phase solver (recorded rotations, back-transformed top-k) against the classic one, 1 and 3 threads
        for (int n : {2, 5, 17, 64, 130, 200}) {
            std::vector<double> B((size_t)n * n), A((size_t)n * n, 0.0), U((size_t)n * n), D(n), Ut((size_t)n * n), Dt(n);
            for (auto& v : B) v = N(g);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0;
                    for (int k = 0; k < n; ++k) s += B[i + (size_t)k * n] * B[j + (size_t)k * n] * std::exp(-0.2 * k);
                    A[i + (size_t)j * n] = s;
                }
            int r = 0;
            nleh::eigen_decomposition(A.data(), n, 0.0, U.data(), D.data(), &r);
            for (int threads : {1, 3}) {
                const int k = n > 2 ? n / 3 : n;
                if (!nleh::sym_eigen_top(A.data(), n, k, threads, Ut.data(), Dt.data())) { ++bad; continue; }
                double ev = 0, res = 0, orth = 0;
                for (int j = 0; j < n; ++j) ev = std::fmax(ev, std::fabs(D[j] - Dt[j]));
                for (int j = 0; j < k; ++j) {
                    for (int i = 0; i < n; ++i) {
                        double s = 0;
                        for (int l = 0; l < n; ++l) s += A[i + (size_t)l * n] * Ut[l + (size_t)j * n];
                        res = std::fmax(res, std::fabs(s - Dt[j] * Ut[i + (size_t)j * n]));
                    }
                    for (int j2 = 0; j2 <= j; ++j2) {
                        double s = 0;
                        for (int i = 0; i < n; ++i) s += Ut[i + (size_t)j * n] * Ut[i + (size_t)j2 * n];
                        orth = std::fmax(orth, std::fabs(s - (j == j2)));
                    }
                }
                const double scale = std::fabs(D[0]) + 1e-300;
                if (ev > 1e-11 * scale || res > 1e-10 * scale || orth > 1e-11) {
                    std::printf("n=%d threads=%d: top-k solver ev %g res %g orth %g\n", n, threads, ev, res, orth);
                    ++bad;
                }
            }
            // the reference's post-processing on the top-k form
            int rt = 0;
            std::vector<double> Uk((size_t)n * n), Dk(n);
            if (!nleh::eigen_decomposition_top(A.data(), n, 1e-10, std::max(1, n / 4), Uk.data(), Dk.data(), &rt) || rt < 1) ++bad;
            for (int j = 1; j < n; ++j) if (Dk[j] > Dk[j - 1] + 1e-12 * std::fabs(Dk[0])) ++bad;
            // and the bisection form of it (Sturm counts; eight shifts a vector, up to eight vectors a sweep), and the
            // few-below-the-cut form the deflated root of Wa uses
            int rb = 0, kept = 0;
            const int kb = std::max(1, n / 4);
            std::vector<double> Ub((size_t)n * n), Db(n), Dlow(n), Ulow((size_t)n * n);
            double lmax = 0, lmin = 0;
            if (!nleh::eigen_decomposition_topk(A.data(), n, 1e-10, kb, Ub.data(), Db.data(), &rb) || rb != rt) { std::printf("n=%d: bisection rank %d vs %d\n", n, rb, rt); ++bad; }
            for (int j = 0; j < std::min(kb, n); ++j) if (std::fabs(Db[j] - Dk[j]) > 1e-11 * (std::fabs(Dk[0]) + 1e-300)) ++bad;
            if (!nleh::sym_eigen_below(A.data(), n, 1e-10, n / 8, &kept, &lmax, &lmin, Dlow.data(), Ulow.data()) || kept != rt) ++bad;
            if (std::fabs(lmax - Dk[0]) > 1e-11 * (std::fabs(Dk[0]) + 1e-300)) ++bad;
        }
        {   // not positive definite: must be refused
            std::vector<double> A = {1, 2, 2, 1}, L(4), Li(4);
            double tr;
            if (nleh::cholesky_with_inverse(A.data(), 2, L.data(), Li.data(), &tr)) { std::printf("indefinite matrix accepted\n"); ++bad; }
        }
        std::printf("bad=%d\n", bad);
        return bad;
    }
""")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_eigensolver_under_asan_ubsan(tmp_path):
    drv = tmp_path / "drv.cpp"
    drv.write_text(DRIVER)
    exe = tmp_path / "drv"
    cmd = ["g++", "-O1", "-g", "-fopenmp-simd", "-pthread", "-Wno-psabi", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", CSRC, str(drv), os.path.join(CSRC, "eigen_sym.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "bad=0" in r.stdout
