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
        std::printf("bad=%d\n", bad);
        return bad;
    }
""")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_eigensolver_under_asan_ubsan(tmp_path):
    drv = tmp_path / "drv.cpp"
    drv.write_text(DRIVER)
    exe = tmp_path / "drv"
    cmd = ["g++", "-O1", "-g", "-fopenmp-simd", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", CSRC, str(drv), os.path.join(CSRC, "eigen_sym.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "bad=0" in r.stdout
