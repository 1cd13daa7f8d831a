"""Device dense solvers (csrc/dense64.hip + devsolve.hip): the symmetric eigen-computation behind `eigenDecomposition`
(reference src/filter.cpp:204-228: lower triangle read, eigenvalues descending, cut at 1e-10) with its O(n^3) part on the GPU,
and the Cholesky factor with inverse -- against LAPACK through numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spectrum_matrix(n, seed, kind):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    if kind == "wa":  # like Wa at the benchmark configs: 1e-5 down to 1e-17, a cluster below the 1e-10 cut
        lam = np.concatenate([np.geomspace(1.4e-5, 1.2e-10, n - n // 9), np.geomspace(8e-11, 1e-17, n // 9)])
    elif kind == "q":  # like Q: (0, 1], top eigenvalue ~ 1
        lam = np.concatenate([[0.99999855], np.linspace(0.92, 1e-3, n - 1) ** 2])
    else:  # indefinite, repeated eigenvalues (the tridiagonal form splits)
        lam = np.concatenate([np.full(n // 4, 0.5), rng.standard_normal(n - n // 4)])
    lam = np.sort(lam)[::-1]
    return (Q * lam) @ Q.T, lam


@pytest.mark.parametrize("n,kind,k", [(3, "q", 3), (17, "mixed", 5), (64, "q", 20), (200, "wa", 30), (257, "mixed", 40),
                                      (400, "q", 50), (800, "q", 100), (900, "wa", 100), (1152, "q", 64)])
def test_device_eigensolver_matches_lapack(nle, ctx, n, kind, k):
    M, lam = _spectrum_matrix(n, 100 + n, kind)
    M = 0.5 * (M + M.T)
    Mlow = np.tril(M) + np.triu(np.full((n, n), 7.0), 1)       # the upper triangle must not be read (:207)
    w = np.linalg.eigvalsh(M)[::-1]
    scale = np.abs(w).max()
    U, D, r = ctx.sym_eigen_device(Mlow, 0, k)
    assert np.all(np.diff(D) <= 8 * np.finfo(float).eps * scale)
    assert np.abs(D - w).max() <= 64 * n * np.finfo(float).eps * scale
    assert r == int(np.sum(np.cumprod(D >= 1e-10)))
    # eigenvectors: residual and orthonormality (signs are arbitrary)
    R = M @ U - U * D[:k]
    assert np.abs(R).max() <= 1e3 * n * np.finfo(float).eps * scale
    assert np.abs(U.T @ U - np.eye(k)).max() <= 1e-11
    # a range in the middle / at the end of the spectrum (what the deflated root of Wa asks for)
    first = max(0, n - max(2, n // 9))
    cnt = n - first
    U2, D2, _ = ctx.sym_eigen_device(Mlow, first, cnt)
    assert np.array_equal(D2, D)
    R2 = M @ U2 - U2 * D[first:first + cnt]
    assert np.abs(R2).max() <= 1e3 * n * np.finfo(float).eps * scale
    assert np.abs(U2.T @ U2 - np.eye(cnt)).max() <= 1e-10


def test_device_eigensolver_rank_cut_matches_host_solver(nle, ctx):
    """the count of eigenvalues >= 1e-10 (src/filter.cpp:213-216) is the host solver's on a Wa-like spectrum"""
    n = 400
    M, lam = _spectrum_matrix(n, 5, "wa")
    M = 0.5 * (M + M.T)
    Uh, Dh, rh = nle.eigen_decomposition_top(M, 1)
    _, Dd, rd = ctx.sym_eigen_device(M, 0, 0)
    assert rd == rh
    assert np.abs(Dd - Dh).max() <= 1e-19


@pytest.mark.parametrize("n", [4, 5, 15, 16, 63, 64, 65, 128, 129, 320, 321, 448, 449, 640, 641, 832, 833, 960, 961])
def test_device_reduction_at_the_switches_of_its_distribution(nle, ctx, n):
    """k_sytrd_wave (csrc/dense64.hip) is instantiated per padded column length (64 x {5, 7, 10, 13, 15, 18} rows), deals
    four columns a wave up to n = 640 and two above, polls with two request sets up to 640 rows, and its lanes own rows
    l, l + 64, ...: the orders on either side of each of those switches, and the smallest ones (one or two workgroups,
    waves that own no column at all), against LAPACK"""
    M, lam = _spectrum_matrix(n, 7000 + n, "q" if n % 2 else "mixed")
    M = 0.5 * (M + M.T)
    w = np.linalg.eigvalsh(M)[::-1]
    scale = np.abs(w).max()
    k = min(n, 8)
    U, D, r = ctx.sym_eigen_device(M, 0, k)
    assert np.abs(D - w).max() <= 64 * n * np.finfo(float).eps * scale
    assert np.abs(M @ U - U * D[:k]).max() <= 1e3 * n * np.finfo(float).eps * scale
    assert np.abs(U.T @ U - np.eye(k)).max() <= 1e-11


@pytest.mark.parametrize("n", [1, 5, 32, 33, 200, 449, 900])
def test_device_cholesky_with_inverse(nle, ctx, n):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 5))
    M = X @ X.T / n + 1e-3 * np.eye(n)
    Mlow = np.tril(M) + np.triu(np.full((n, n), -3.0), 1)
    L, Li, tr, ok = ctx.cholesky_device(Mlow)
    assert ok
    Lref = np.linalg.cholesky(M)
    assert np.abs(L - Lref).max() <= 1e-12 * np.abs(Lref).max()
    assert np.abs(np.triu(L, 1)).max() == 0.0 and np.abs(np.triu(Li, 1)).max() == 0.0
    assert np.abs(Li @ L - np.eye(n)).max() <= 1e-10
    assert abs(tr - np.trace(np.linalg.inv(M))) <= 1e-9 * tr


def test_device_cholesky_reports_an_indefinite_matrix(nle, ctx):
    n = 100
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, n))
    M = X + X.T
    _, _, _, ok = ctx.cholesky_device(M)
    assert not ok


def test_a_device_that_cannot_run_the_persistent_reduction_falls_back_to_the_host_solver(nle, oracle, ctx):
    """the Householder reduction is ONE persistent launch whose workgroups must all be resident; a device with
    fewer compute units than workgroups (a partition, a smaller part) or a timed-out hand-off used to fail the whole train
    with NLE_ERR_NUMERIC although the host solver was there.  NLE_SYTRD_G=250 asks for more workgroups than the chip offers
    (+ the 16 held back): the train must take the host solvers and give what NLE_HOST_WA / NLE_HOST_Q give (to rounding: the
    matrices handed to the host solver were formed on the device on one route, on the host on the other); the stand-alone
    entry point, which has no host form to fall back on, reports the reason."""
    import os
    H, W, nr, nc, hx, hy, T, K, L = 192, 256, 17, 18, 48.0, 30.0, 6, 24, 4          # 306 samples: above the device threshold
    x = oracle.synthetic_luminance(H, W).astype(np.float32)

    def run(env):
        os.environ.update(env)
        try:
            f = nle.NLEFilter(ctx).train_filter(x, nr, nc, hx, hy, T, K)
            ev, Y = f.eigvals.copy(), f.apply_layers(x, L).cpu().numpy()
            f.close()
            return ev, Y
        finally:
            for k in env:
                del os.environ[k]

    ev_h, Y_h = run({"NLE_HOST_WA": "1", "NLE_HOST_Q": "1"})
    ev_f, Y_f = run({"NLE_SYTRD_G": "250"})
    assert np.abs(ev_f - ev_h).max() <= 1e-12 and np.abs(Y_f - Y_h).max() <= 1e-6 * np.abs(Y_h).max()
    ev_d, Y_d = run({})                                                         # and the device solvers agree to rounding
    assert np.abs(ev_d - ev_h).max() <= 1e-10 and np.abs(Y_d - Y_h).max() <= 1e-3 * np.abs(Y_h).max()
    os.environ["NLE_SYTRD_G"] = "250"
    try:
        M, _ = _spectrum_matrix(300, 1, "q")
        with pytest.raises(nle.NLEError, match="compute units"):
            ctx.sym_eigen_device(0.5 * (M + M.T), 0, 4)
    finally:
        del os.environ["NLE_SYTRD_G"]
