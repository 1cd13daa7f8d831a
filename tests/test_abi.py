"""The C-ABI library builds, loads, and exports every symbol include/nle.h declares; the
host-only entry points work without a GPU; compute entry points fail loudly without one."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nle.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nle_[a-z_0-9]+)\s*\(", text)) - {"nle_allreduce_fn"})


def test_every_declared_symbol_is_exported_and_bound(nle):
    import ctypes
    lib = nle.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nle.h but not exported"
        ctypes.cast(getattr(lib, name), ctypes.c_void_p)
    assert sorted(nle.EXPORTED_SYMBOLS) == declared, "Python mirror and header disagree"


def test_ctypes_mirror_is_generated_from_the_header():
    """nonlocal-image-edit_amd/_abi.py (signatures + constants of the Python mirror) is what tools/gen_ctypes.py makes of
    include/nle.h today"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ctypes.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_is_gfx950_native(nle):
    """the shared object carries a gfx950 code object (hipcc --offload-arch=gfx950)"""
    blob = open(nle.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for k in (b"k_affinity", b"k_rowpass", b"k_tsgemm", b"k_gram"):
        assert k in blob, k


# (H, W, nRow, nCol) -> expected selected rows / cols, from src/filter.cpp:56-71 evaluated by hand
GRID_CASES = [
    (267, 400, 10, 20, list(range(16, 267, 26))[:10], list(range(9, 400, 20))[:20]),
    (512, 512, 10, 20, [26 + 51 * i for i in range(10)], [18 + 25 * i for i in range(20)]),
    (4096, 4096, 20, 10, [109 + 204 * i for i in range(20)], [207 + 409 * i for i in range(10)]),
    (8192, 8192, 30, 30, [137 + 273 * i for i in range(30)], [137 + 273 * i for i in range(30)]),
]


@pytest.mark.parametrize("H,W,nr,nc,rows,cols", GRID_CASES)
def test_sample_grid_closed_form(nle, oracle, H, W, nr, nc, rows, cols):
    g = nle.sample_grid(H, W, nr, nc)
    got_r = [g["row_off"] + i * g["row_step"] for i in range(g["n_sel_rows"])]
    got_c = [g["col_off"] + i * g["col_step"] for i in range(g["n_sel_cols"])]
    assert got_r == rows and got_c == cols
    sr, sc = oracle.sample_grid(H, W, nr, nc)
    assert got_r == sr.tolist() and got_c == sc.tolist()


def _brute_force_selected(H, W, nr, nc):
    """literal double loop of samplePixels, src/filter.cpp:56-80"""
    rs, cs = H // nr, W // nc
    ro, co = (rs - 1 + (H - rs * nr)) // 2, (cs - 1 + (W - cs * nc)) // 2
    sel = []
    for r in range(H):
        for c in range(W):
            if r >= ro and c >= co and r <= H - ro and c <= W - co and (r - ro) % rs == 0 and (c - co) % cs == 0:
                sel.append(r * W + c)
    return sel


@pytest.mark.parametrize("H,W,nr,nc", [(15, 20, 10, 7), (7, 9, 7, 9), (33, 47, 3, 4), (10, 10, 1, 1), (5, 64, 5, 3),
                                        (12, 12, 5, 5), (9, 31, 4, 30)])
def test_sample_grid_matches_literal_scan(nle, oracle, H, W, nr, nc):
    """includes step == 1 cases where the realised count exceeds nRow*nCol"""
    want = _brute_force_selected(H, W, nr, nc)
    g = nle.sample_grid(H, W, nr, nc)
    got = [(g["row_off"] + i * g["row_step"]) * W + g["col_off"] + j * g["col_step"]
           for i in range(g["n_sel_rows"]) for j in range(g["n_sel_cols"])]
    assert got == want
    sel, rest = oracle.sample_pixels(H, W, nr, nc)
    assert sel.tolist() == want and sel.size + rest.size == H * W


def test_too_many_samples_is_an_error(nle, oracle):
    with pytest.raises(nle.NLEError):
        nle.sample_grid(10, 10, 11, 2)
    with pytest.raises(RuntimeError, match="Number of samples per row and col must be <= that of image"):
        oracle.compute_kernel(np.zeros((10, 10)), 11, 2, 1.0, 1.0)


def test_slab_rows_cover_image(nle, oracle):
    for H in (1, 7, 267, 4096):
        for G in (1, 2, 3, 8):
            prev = 0
            for g in range(G):
                r0, r1 = nle.slab_rows(H, g, G)
                assert (r0, r1) == oracle.slab_rows(H, g, G)
                assert r0 == prev and r1 >= r0
                prev = r1
            assert prev == H


def test_eigensolver_against_lapack(nle):
    rng = np.random.default_rng(11)
    for n in (1, 2, 17, 64, 200):
        A = rng.standard_normal((n, n))
        A = A @ A.T / n + 1e-3 * np.eye(n)
        U, D = nle.eigen_decomposition(A)
        w = np.linalg.eigvalsh(A)[::-1]
        assert D.size == n and np.allclose(D, w, rtol=1e-12, atol=1e-13)
        assert np.abs(U.T @ U - np.eye(n)).max() < 1e-12
        assert np.abs(U @ np.diag(D) @ U.T - A).max() < 1e-12 * max(1.0, np.abs(A).max()) * n


def test_compute_needs_a_gpu_and_says_so(nle):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nle.Context(0)
    # straight through the C ABI: status NLE_ERR_HIP, never a silent CPU path
    import ctypes as C
    h = C.c_void_p()
    st = nle.lib().nle_ctx_create(0, None, C.byref(h))
    assert st == nle.NLE_ERR_HIP and not h.value
    assert b"HIP" in nle.lib().nle_last_error(None) or b"hip" in nle.lib().nle_last_error(None)


def test_rccl_bootstrap_argument_checks_need_no_gpu(nle):
    """the communicator bootstrap refuses malformed calls before it loads librccl or touches a device"""
    import ctypes as C
    L = nle.lib()
    short = (C.c_char * 64)()
    assert L.nle_rccl_unique_id(None, 128) == nle.NLE_ERR_INVALID
    assert L.nle_rccl_unique_id(short, 64) == nle.NLE_ERR_INVALID
    assert L.nle_ctx_init_rccl(None, 0, 1, short, 128) == nle.NLE_ERR_INVALID
    assert L.nle_ctx_set_rccl_comm(None, 0, 1, None) == nle.NLE_ERR_INVALID


def test_lab8_tables_are_the_oracles(nle, oracle):
    """the fixed-point tables behind nle_bgr2lab8 and the C++ surface's bgr2lab8 (OpenCV's 8-bit BGR -> Lab) are, entry for
    entry, the ones the oracle builds: the integer arithmetic on top of them then cannot differ"""
    g, c, k = nle.lab8_tables()
    go, co, ko = oracle.lab8_tables()
    assert np.array_equal(g, go) and np.array_equal(c, co) and np.array_equal(k, ko)
    assert g[0] == 0 and g[255] == 2040 and c.size == 3072 and int(k[1].sum()) == 4096
    # made in single precision with OpenCV's truncating cube root: exactly two entries differ (by one) from the table a
    # double-precision cube root gives -- what decides the last 16 pixels of the reference's flower-filtered.png
    t = np.arange(3072) / 2040.0
    c64 = np.rint(32768 * np.where(t < 216 / 24389, t * (841 / 108) + 16 / 116, np.cbrt(t))).astype(np.int64)
    assert np.flatnonzero(c64 != c).tolist() == [49, 628] and int(c[49]) == c64[49] - 1 and int(c[628]) == c64[628] + 1


def test_lab8_inverse_tables_are_the_oracles(nle, oracle):
    """the same for the integer Lab -> BGR (OpenCV's Lab2RGBinteger) behind nle_lab2bgr8 and the C++ surface's lab2bgr8"""
    yf, ab, ig, k = nle.lab8_inverse_tables()
    yfo, abo, igo, ko = oracle.lab8_inverse_tables()
    assert np.array_equal(yf, yfo) and np.array_equal(ab, abo) and np.array_equal(ig, igo) and np.array_equal(k, ko)
    assert ig[0] == 0 and ig[-1] == 255 and yf[255, 0] == 16384 and yf[255, 1] == 16384 and yf[0, 1] == 2260
    assert ab[0] == -1335 and ab[3390 + 8145] == 145 and ab[16384 + 8145] == 16384       # both branches of f^-1, C division


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nonlocal-image-edit_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "nle_oracle" not in text and "oracle/" not in text.replace("Nothing here imports `oracle/`", ""), fn


def test_synthetic_generator_matches_oracle(oracle):
    import __graft_entry__ as entry
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    a = synth.synthetic_luminance(37, 53)
    assert np.array_equal(a, oracle.synthetic_luminance(37, 53))
    assert np.array_equal(synth.synthetic_luminance(37, 53, rows=(5, 20)), a[5:20])
    assert a.min() >= 0 and a.max() <= 255 and np.array_equal(a, np.rint(a))


def test_topk_lanczos_matches_the_full_solver(nle, oracle):
    """nle_topk_eigen_decomposition = the reference's USE_SPECTRA solver for Q (src/filter.cpp:170-199):
    nev = min(K, n - 1) pairs of largest magnitude, tolerance 1e-10, sorted descending, cut at eps.  Against the full
    solver (LAPACK) on matrices shaped like Q: eigenvalues in (0, 1] decaying towards a cluster at 0, a few tiny
    negative ones, and the slight asymmetry Wa gives Q (the FULL matrix is multiplied, like Spectra's DenseGenMatProd)."""
    rng = np.random.default_rng(3)
    for n, K in ((40, 5), (200, 50), (200, 10), (333, 100), (64, 200), (12, 11)):
        X = np.linalg.qr(rng.standard_normal((n, n)))[0]
        lam = np.concatenate([[1.0], 0.95 * 0.85 ** np.arange(n - 1)])
        lam[n // 2:] *= 1e-6
        lam[-3:] = [-1e-7, -2e-8, -1e-12]
        Q = (X * lam) @ X.T
        Q = Q + 1e-13 * np.triu(rng.standard_normal((n, n)), 1)       # not symmetric, like Wa + S (..) S
        w, v = np.linalg.eigh((Q + Q.T) / 2)
        w, v = w[::-1], v[:, ::-1]
        U, D = nle.topk_eigen_decomposition(Q, K)
        nev = min(K, n - 1)
        want = np.sort(w[np.argsort(-np.abs(w))[:nev]])[::-1]
        want = want[:np.argmax(want < 1e-10)] if (want < 1e-10).any() else want
        assert D.size == want.size, (n, K, D.size, want.size)
        assert np.all(np.diff(D) <= 0)
        assert np.abs(D - want).max() < 1e-8
        assert np.abs(U.T @ U - np.eye(D.size)).max() < 1e-8
        # eigenvectors: same invariant subspaces (compare projectors on the well separated leading block)
        kk = min(D.size, 5)
        assert np.abs(U[:, :kk] @ U[:, :kk].T - v[:, :kk] @ v[:, :kk].T).max() < 1e-6


def test_leading_eigenpairs_by_bisection(nle):
    """nle_eigen_decomposition_topk: what the train path computes of Q on the host -- the K largest eigenvalues by bisection
    on Sturm counts of the tridiagonal form, the rank of the 1e-10 cut (src/filter.cpp:214) as one more count, the
    eigenvectors by inverse iteration.  Against LAPACK and against the QL form (nle_eigen_decomposition_top) on spectra
    that cross the cut, cluster, repeat, and on the degenerate matrices."""
    rng = np.random.default_rng(19)
    cases = []
    for n, k in ((200, 50), (196, 50), (64, 32), (333, 100), (30, 3), (16, 8), (9, 1)):
        lam = np.concatenate([np.geomspace(1.5, 1.2e-10, n - n // 10), np.geomspace(9e-11, 1e-14, n // 10)])
        cases.append(("through the cut", n, k, lam))
    lam = np.sort(np.concatenate([1.0 - 1e-9 * np.arange(6), 0.5 + 1e-12 * np.arange(5), rng.uniform(0, 0.4, 109)]))[::-1]
    cases.append(("clusters", 120, 40, lam))
    lam = np.sort(np.concatenate([[0.9] * 4, [0.7] * 3, rng.uniform(0, 0.5, 93)]))[::-1]
    cases.append(("repeated", 100, 20, lam))
    cases.append(("wa", 200, 60, 1.5e-5 * 0.9 ** np.arange(200)))
    cases.append(("scaled up", 80, 20, 3e7 * 0.8 ** np.arange(80)))
    cases.append(("indefinite", 90, 30, np.linspace(2.0, -1.0, 90)))
    for name, n, k, lam in cases:
        X = np.linalg.qr(rng.standard_normal((n, n)))[0]
        A = (X * lam) @ X.T
        A = (A + A.T) / 2
        U, Dk, r = nle.eigen_decomposition_topk(A, k)
        Uq, Dq, rq = nle.eigen_decomposition_top(A, k)
        w = np.linalg.eigvalsh(A)[::-1]
        scale = np.abs(w).max()
        # the count of the cut: LAPACK's, unless an eigenvalue sits within rounding of 1e-10 (none of these spectra does)
        assert r == rq == int((w >= 1e-10).sum()), (name, n, r, rq)
        assert Dk.shape == (k,) and np.all(np.diff(Dk) <= 0)
        assert np.abs(Dk - w[:k]).max() < 1e-13 * scale * n, (name, np.abs(Dk - w[:k]).max())
        assert np.abs(Dk - Dq[:k]).max() < 1e-13 * scale * n
        assert U.shape == (n, k)
        assert np.abs(U.T @ U - np.eye(k)).max() < 1e-10, (name, np.abs(U.T @ U - np.eye(k)).max())
        res = np.abs(A @ U - U * Dk).max()
        assert res < 1e-12 * scale * n, (name, res)
    # 2 kmax > n: the QL form, same contract
    A = np.diag(np.linspace(1.0, 0.01, 50))
    U, Dk, r = nle.eigen_decomposition_topk(A, 40)
    assert r == 50 and np.allclose(Dk, np.linspace(1.0, 0.01, 50)[:40], atol=1e-14)
    # diagonal (every off-diagonal of T is zero), identity (all eigenvalues equal), zero matrix (no eigenvalue kept)
    U, Dk, r = nle.eigen_decomposition_topk(A, 7)
    assert r == 50 and np.abs(np.abs(U[:7, :7]) - np.eye(7)).max() < 1e-12
    U, Dk, r = nle.eigen_decomposition_topk(np.eye(40), 5)
    assert r == 40 and np.allclose(Dk, 1.0, atol=1e-15) and np.abs(U.T @ U - np.eye(5)).max() < 1e-12
    U, Dk, r = nle.eigen_decomposition_topk(np.zeros((40, 40)), 5)
    assert r == 0 and np.abs(Dk).max() < 1e-300


def test_leading_eigenvectors_by_inverse_iteration(nle):
    """nle_eigen_decomposition_top with kmax <= n / 2 takes the eigenvectors from inverse iteration on the tridiagonal
    form (csrc/eigen_sym.cpp): against LAPACK on spectra like Q's (decaying from 1), with tight clusters, with exactly
    repeated eigenvalues (the tridiagonal splits), on a diagonal matrix, and with the spectrum of a truncated Wa."""
    rng = np.random.default_rng(9)
    cases = []
    for n, k in ((200, 50), (200, 10), (64, 32), (333, 100), (30, 3)):
        lam = np.concatenate([[1.0], 0.97 * 0.9 ** np.arange(n - 1)])
        cases.append(("decay", n, k, lam))
    n = 120
    lam = np.sort(np.concatenate([1.0 - 1e-9 * np.arange(6), 0.5 + 1e-12 * np.arange(5), rng.uniform(0, 0.4, n - 11)]))[::-1]
    cases.append(("clusters", n, 40, lam))
    lam = np.sort(np.concatenate([[0.9] * 4, [0.7] * 3, rng.uniform(0, 0.5, 93)]))[::-1]
    cases.append(("repeated", 100, 20, lam))
    cases.append(("wa", 200, 60, 1.5e-5 * 0.9 ** np.arange(200)))
    # enough vectors in enough separate clusters that the clusters are dealt to threads (csrc/eigen_sym.cpp)
    lam = np.sort(np.concatenate([np.linspace(1.0, 0.3, 60), 0.25 - 1e-9 * np.arange(30), rng.uniform(0, 0.2, 810)]))[::-1]
    cases.append(("threads", 900, 100, lam))
    # one cluster of 100 (60 eigenvalues 1e-12 apart, 40 exactly repeated): block inverse iteration + Cholesky-QR, two passes
    lam = np.sort(np.concatenate([1.0 - 1e-10 * rng.uniform(0, 1, 60), np.full(40, 1.0 - 2e-10), rng.uniform(0, 0.9, 800)]))[::-1]
    cases.append(("block cluster", 900, 120, lam))
    for name, n, k, lam in cases:
        X = np.linalg.qr(rng.standard_normal((n, n)))[0]
        A = (X * lam) @ X.T
        A = (A + A.T) / 2
        U, D, r = nle.eigen_decomposition_top(A, k)
        w = np.linalg.eigvalsh(A)[::-1]
        scale = np.abs(w).max()
        assert np.abs(D - w).max() < 1e-13 * scale * n, name
        assert U.shape == (n, k)
        assert np.abs(U.T @ U - np.eye(k)).max() < 1e-10, (name, np.abs(U.T @ U - np.eye(k)).max())
        res = np.abs(A @ U - U * D[:k]).max()
        assert res < 1e-12 * scale * n, (name, res)
    A = np.diag(np.linspace(1.0, 0.01, 50))
    U, D, r = nle.eigen_decomposition_top(A, 7)
    assert np.abs(np.abs(U[:7, :7]) - np.eye(7)).max() < 1e-12
