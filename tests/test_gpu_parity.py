"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs (sizes the oracle finishes in seconds).

Tolerances:
  * BASELINE.json north_star bar: <= 1e-4 relative L2 per detail layer  -> PER_LAYER_TOL.  It is asserted, with no
    slack, on the two formulations auto mode selects (the fp64 table form, NLE_MODE_PHI_FREE on integer-valued planes,
    and the fp64 literal decomposition, NLE_MODE_MATERIALISED_F64) on every case.  The two fp32 formulations
    (NLE_MODE_MATERIALISED, NLE_MODE_PHI_FREE_EXP) are opt-in: the same bar on ordinary cases, 5e-4 on the one case
    their own selection rule would have excluded (fewer than 64 pixels per sample).
  * kernels checked in isolation against fp64 numpy on the SAME fp32 inputs: 1e-5 (fp32
    MFMA accumulation) or 1e-9 (fp64 reductions)
"""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

PER_LAYER_TOL = 1e-4  # BASELINE.json: "within 1e-4 relative L2 per detail layer"

# (H, W, nRow, nCol, hx, hy, T, K, L)
SMALL_CASES = [
    (48, 64, 4, 5, 16.0, 30.0, 10, 8, 4),
    (96, 128, 6, 8, 32.0, 30.0, 10, 10, 4),
    (15, 20, 10, 7, 8.0, 30.0, 5, 6, 3),     # step == 1: realised p = 96 > 10*7
    (33, 47, 3, 4, 20.0, 25.0, 1, 4, 2),     # ragged sizes, a single Sinkhorn iteration
    (64, 64, 8, 8, 16.0, 30.0, 7, 70, 5),    # K larger than the kept spectrum of Q
]


def _torch():
    import torch
    return torch


def _sel_pixels(oracle, H, W, nr, nc):
    sr, sc = oracle.sample_grid(H, W, nr, nc)
    return (np.repeat(sr, sc.size) * W + np.tile(sc, sr.size)).astype(np.int64)


def _align_signs(A, B):
    """flip columns of A to match B (eigenvector signs are arbitrary)."""
    s = np.sign(np.sum(A * B, axis=0))
    s[s == 0] = 1.0
    return A * s[None, :]


@pytest.mark.parametrize("case", SMALL_CASES[:4])
def test_compute_kernel_matches_oracle(nle, oracle, ctx, case):
    H, W, nr, nc, hx, hy, *_ = case
    x = oracle.synthetic_luminance(H, W)
    perm, Ka_o, Kab_o = oracle.compute_kernel(x, nr, nc, hx, hy)
    Ka, kab = ctx.compute_kernel(x.astype(np.float32), nr, nc, hx, hy)
    p = Ka_o.shape[0]
    assert Ka.shape == (p, p)
    assert np.abs(Ka - Ka_o).max() < 1e-13
    kab = kab.cpu().numpy()
    assert kab.shape == (H * W, nle.ld(p))
    # oracle Kab is p x (N-p) in [rest] order; device rows are ALL pixels, natural order
    full = np.empty((H * W, p))
    full[perm[:p]] = Ka_o.T
    full[perm[p:]] = Kab_o.T
    assert np.abs(kab[:, :p] - full).max() < 2e-6          # fp32 exp of values in [0, 1]
    assert np.all(kab[:, p:] == 0)


@pytest.mark.parametrize("case", SMALL_CASES[:4])
def test_nystrom_matches_oracle(nle, oracle, ctx, case):
    H, W, nr, nc, hx, hy, *_ = case
    x = oracle.synthetic_luminance(H, W)
    perm, Ka_o, Kab_o = oracle.compute_kernel(x, nr, nc, hx, hy)
    lam_o, phi_o = oracle.nystrom_approximation(Ka_o, Kab_o)
    lam, phi, r = ctx.nystrom(x.astype(np.float32), nr, nc, hx, hy)
    assert r == lam_o.size
    assert rel_l2(lam, lam_o) < 1e-10
    phi = phi.cpu().numpy()[:, :r].astype(np.float64)
    nat = np.empty_like(phi_o)
    nat[perm] = phi_o
    phi = _align_signs(phi, nat)
    # columns scale like 1/lambda_k; compare in the lambda-weighted metric the path uses them in
    assert rel_l2(phi * lam_o[None, :], nat * lam_o[None, :]) < 1e-4
    # sample rows are the exact V_A rows (fp32-rounded)
    sel = perm[:Ka_o.shape[0]]
    assert np.abs(phi[sel] - nat[sel]).max() < 1e-6


@pytest.mark.parametrize("M,kd,nc", [(1000, 20, 12), (777, 96, 96), (4099, 200, 50), (300, 7, 260), (129, 300, 33)])
def test_ts_gemm_matches_numpy(nle, ctx, M, kd, nc):
    torch = _torch()
    rng = np.random.default_rng(M + kd)
    lda = nle.ld(kd)
    A = np.zeros((M, lda), dtype=np.float32)
    A[:, :kd] = rng.standard_normal((M, kd)).astype(np.float32)
    B = rng.standard_normal((kd, nc))
    Cd = ctx.ts_gemm(torch.as_tensor(A, device="cuda"), kd, B).cpu().numpy()
    ref = A[:, :kd].astype(np.float64) @ B.astype(np.float32).astype(np.float64)
    assert Cd.shape == (M, nle.ld(nc))
    assert rel_l2(Cd[:, :nc], ref) < 1e-5
    assert np.all(Cd[:, nc:] == 0)


@pytest.mark.parametrize("M,r", [(5, 3), (2, 2), (1000, 20), (5000, 96), (3001, 200), (2000, 400), (1500, 900), (700, 1100)])
def test_sinkhorn_gram_rowscale_kernels(nle, oracle, ctx, M, r):
    """Each N-sized kernel against fp64 numpy on the SAME fp32 matrix."""
    torch = _torch()
    rng = np.random.default_rng(r)
    ldp = nle.ld(r)
    # a positive, well-conditioned "phi": rows of an orthonormal-ish basis plus offset
    phi = np.zeros((M, ldp), dtype=np.float32)
    phi[:, :r] = (np.abs(rng.standard_normal((M, r))) + 0.1).astype(np.float32) / np.sqrt(r)
    lam = np.sort(rng.uniform(0.5, 2.0, r))[::-1].copy()
    d_phi = torch.as_tensor(phi, device="cuda")
    P = phi[:, :r].astype(np.float64)
    T = 4
    uc, ur = ctx.sinkhorn_scalings(d_phi, r, lam, T)
    rv = np.ones(M)
    for _ in range(T):
        uc_ref = lam * (P.T @ rv)
        c, _ = oracle.inplace_reciprocal(P @ uc_ref)
        ur_ref = lam * (P.T @ c)
        rv, _ = oracle.inplace_reciprocal(P @ ur_ref)
    assert rel_l2(uc, uc_ref) < 1e-10
    assert rel_l2(ur, ur_ref) < 1e-10
    cs = ctx.row_scalings(d_phi, r, uc_ref).cpu().numpy()
    c_ref, _ = oracle.inplace_reciprocal(P @ uc_ref)
    assert rel_l2(cs, c_ref) < 1e-12
    G = ctx.gram(d_phi, r, uc_ref)
    Z = P * c_ref[:, None]
    assert rel_l2(G, Z.T @ Z) < 1e-5
    assert np.abs(G - G.T).max() == 0.0


def test_reciprocal_zeroing(nle, oracle, ctx):
    """inplaceReciprocal semantics (src/filter.cpp:42-54): |v| < eps -> 0, not inf."""
    torch = _torch()
    phi = np.zeros((8, 4), dtype=np.float32)
    phi[:4, :2] = np.array([[1, 2], [0, 0], [3, -3], [1e-6, 0]], dtype=np.float32)
    u = np.array([1.0, 1.0])
    out = ctx.row_scalings(torch.as_tensor(phi, device="cuda"), 2, u).cpu().numpy()
    ref, _ = oracle.inplace_reciprocal(phi[:, :2].astype(np.float64) @ u)
    assert np.array_equal(out, ref)
    assert out[1] == 0.0 and out[2] == 0.0 and np.isfinite(out).all()


def _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L):
    f = nle.NLEFilter(ctx).train_filter(x.astype(np.float32), nr, nc, hx, hy, T, K)
    Y = f.apply_layers(x.astype(np.float32), L).cpu().numpy().astype(np.float64)
    return f, Y


FP64_MODES = (2, 4, 5)   # what auto mode resolves to: bar asserted without slack


@pytest.fixture(params=[2, 4, 5, 1, 3], ids=["tables_f64", "materialised_f64", "streamed_f64", "materialised_f32", "phi_free_exp_f32"])
def mode(request, ctx):
    """run the test under every formulation of the N-sized passes (NLE_MODE_* in include/nle.h)"""
    ctx.set_mode(request.param)
    yield request.param
    ctx.set_mode(0)


@pytest.mark.parametrize("case", SMALL_CASES)
def test_train_apply_layers_match_oracle(nle, oracle, ctx, mode, case):
    H, W, nr, nc, hx, hy, T, K, L = case
    x = oracle.synthetic_luminance(H, W)
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f, Y = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    info = f.info()
    assert info["K"] == S_o.size and info["n_local"] == H * W
    # 15x20 with 96 samples (a third of the pixels are samples, lambda_min(Ka) = 2e-5) is outside
    # what the Phi-free formulation is selected for in auto mode (< 64 pixels per sample); forced, it
    # still lands within 5e-4
    g = nle.sample_grid(H, W, nr, nc)
    forced_tiny = mode not in FP64_MODES and H * W < 64 * g["n_sel_rows"] * g["n_sel_cols"]
    tol = 5e-4 if forced_tiny else PER_LAYER_TOL
    assert f.diag()["formulation"] == mode
    assert rel_l2(f.eigvals, S_o) < (5e-5 if forced_tiny else 1e-5)
    for j in range(L):
        assert rel_l2(Y[j], Y_o[j]) < tol, f"layer {j}"
    # V itself, up to per-column sign
    V = f.eigvecs().cpu().numpy()[:, :S_o.size].astype(np.float64)
    ev = rel_l2(_align_signs(V, V_o), V_o)
    print(f"V vs oracle, mode {mode}, case {case[:4]}: {ev:.2e}")
    # fp64 formulations: the fp32 storage of V is what is left (measured 2.5e-8 .. 5.2e-7); the opt-in fp32 ones 3e-7 .. 2.3e-4
    assert ev < (2e-6 if mode in FP64_MODES else 1e-3)
    # weighted sum == NLEFilter::apply with transformEigenValues
    w = [2.0, 3.0, 4.0, 1.0, 0.5][:L]
    y = f.apply(x.astype(np.float32), nle.transform_eigenvalues(f.eigvals, w)).cpu().numpy()
    y_o = oracle.apply_filter(V_o, x, oracle.transform_eigenvalues(S_o, w)).ravel()
    assert rel_l2(y, y_o) < tol


def test_rank_truncated_Ka(nle, oracle, ctx, mode):
    """Ka numerically rank deficient with a clear gap at the 1e-10 cut (4 grey levels, spatial
    bandwidth so wide that pixels of one level are near duplicates): r = 4 < p = 30, and the
    A block of sinkhorn/orthogonalize is the first r samples (src/filter.cpp:247)."""
    H, W = 40, 52
    levels = np.array([60.0, 120.0, 200.0, 90.0])
    x = levels[np.random.default_rng(5).integers(0, 4, size=(H, W))]
    nr, nc, hx, hy, T, K, L = 5, 6, 1e8, 50.0, 6, 5, 3
    V_o, S_o, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
    p, r = inter["Ka"].shape[0], inter["lam"].size
    assert p == 30 and r == 4, "test input must truncate"
    w_all = np.linalg.eigvalsh(inter["Ka"])[::-1]
    assert inter["lam"][-1] > 1e-2 and abs(w_all[r]) < 1e-11, "cut must not be borderline"
    assert S_o.size == 4  # K' = min(K, kept) = 4 < K
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f, Y = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    assert f.info()["r"] == r and f.info()["p"] == p and f.info()["K"] == 4
    assert rel_l2(f.eigvals, S_o) < 1e-5
    for j in range(L):
        assert rel_l2(Y[j], Y_o[j]) < PER_LAYER_TOL, f"layer {j}"


@pytest.mark.parametrize("case", [SMALL_CASES[1], SMALL_CASES[-1]])
def test_cholesky_and_eigen_forms_of_Ka_agree(nle, oracle, ctx, case, monkeypatch):
    """Full-rank Ka: the Phi-free path factors Ka (and Wa when the 1e-10 cut removes nothing) by Cholesky
    instead of the eigensolver (pipeline.hip solve_Ka / ortho_ss_prepare); NLE_FORCE_EIG=1 keeps the
    eigenpairs.  Both must match the oracle, and each other far below the parity bar."""
    H, W, nr, nc, hx, hy, T, K, L = case
    x = oracle.synthetic_luminance(H, W)
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    ctx.set_mode(2)
    try:
        f_c, Y_c = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
        monkeypatch.setenv("NLE_FORCE_EIG", "1")
        f_e, Y_e = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    finally:
        ctx.set_mode(0)
    assert f_c.info()["K"] == f_e.info()["K"] == S_o.size
    assert rel_l2(f_c.eigvals, f_e.eigvals) < 1e-8
    for j in range(L):
        assert rel_l2(Y_c[j], Y_o[j]) < PER_LAYER_TOL and rel_l2(Y_e[j], Y_o[j]) < PER_LAYER_TOL, f"layer {j}"
        assert rel_l2(Y_c[j], Y_e[j]) < 1e-5, f"layer {j}"


def test_eigvec_range_without_materialising_V(nle, oracle, ctx):
    """the `Eigvec i ... minCoeff ... maxCoeff` banner (src/filter.cpp:506) needs the range of a few columns: on the
    default path they are projected into a temporary, and agree with the materialised matrix"""
    H, W = 96, 128
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    f = nle.NLEFilter(ctx).train_filter(x, 6, 8, 32.0, 30.0, 10, 10)
    mn, mx = f.eigvec_range(4)
    V = f.eigvecs().cpu().numpy()[:, :4].astype(np.float64)      # materialises
    assert np.allclose(mn, V.min(0), rtol=0, atol=1e-6) and np.allclose(mx, V.max(0), rtol=0, atol=1e-6)
    mn2, mx2 = f.eigvec_range(4)                                  # now from the stored matrix
    assert np.allclose(mn2, V.min(0), rtol=0, atol=1e-7) and np.allclose(mx2, V.max(0), rtol=0, atol=1e-7)
    with pytest.raises(nle.NLEError):
        f.eigvec_range(f.info()["K"] + 1)


def test_errors_mirror_reference(nle, oracle, ctx):
    x = oracle.synthetic_luminance(20, 30).astype(np.float32)
    with pytest.raises(nle.NLEError, match="Number of samples per row and col must be <= that of image"):
        nle.NLEFilter(ctx).train_filter(x, 21, 5, 10, 30, 5, 5)  # src/filter.cpp:117-119
    with pytest.raises(nle.NLEError, match="Number of samples per row and col must be <= that of image"):
        ctx.compute_kernel(x, 5, 31, 10, 30)
    f = nle.NLEFilter(ctx).train_filter(x, 4, 5, 10, 30, 5, 5)
    with pytest.raises(nle.NLEError, match="Number of values in channel must match that of training image"):
        f.apply(np.zeros((10, 10), dtype=np.float32), np.ones(f.info()["K"]))  # :447-449
    with pytest.raises(nle.NLEError):
        nle.NLEFilter(ctx).train_filter(x, 4, 5, 10, 30, 0, 5)


def test_apply_properties_small(nle, oracle, ctx, mode):
    """apply is linear; with fS = 1 it is the orthogonal projector onto span(V) (idempotent)."""
    H, W = 64, 80
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    f = nle.NLEFilter(ctx).train_filter(x, 5, 6, 20, 30, 10, 12)
    K = f.info()["K"]
    ones = np.ones(K)
    p1 = f.apply(x, ones).cpu().numpy().reshape(H, W)
    p2 = f.apply(p1, ones).cpu().numpy().reshape(H, W)
    assert rel_l2(p2, p1) < 1e-5
    layers = f.apply_layers(x, 4).cpu().numpy()
    assert rel_l2(layers.sum(0), p1.ravel()) < 1e-5      # responses telescope to 1
    z = oracle.synthetic_luminance(H, W, seed=7).astype(np.float32)
    fs = np.linspace(2.0, 0.5, K)
    lhs = f.apply(2.0 * x - 3.0 * z, fs).cpu().numpy()
    rhs = 2.0 * f.apply(x, fs).cpu().numpy() - 3.0 * f.apply(z, fs).cpu().numpy()
    assert rel_l2(lhs, rhs) < 1e-5


def test_cfg2_both_modes_match_oracle(nle, oracle, ctx):
    """BASELINE.json configs[1]: 512x512, 10x20 samples, K=10, 4 layers, fp32 -- full size against the
    streaming oracle (validated against the literal one in tests/test_golden.py)."""
    H = W = 512
    x = oracle.synthetic_luminance(H, W)
    V_o, S_o = oracle.train_filter_streaming(x, 10, 20, W / 4.0, 30.0, 10, 10)
    Y_o = oracle.apply_layers_streaming(V_o, S_o, x, 4)
    for m in (1, 2, 3):
        ctx.set_mode(m)
        try:
            f, Y = _run_device(nle, ctx, x, 10, 20, W / 4.0, 30.0, 10, 10, 4)
        finally:
            ctx.set_mode(0)
        assert f.info()["r"] == 200
        assert rel_l2(f.eigvals, S_o) < 1e-5
        for j in range(4):
            assert rel_l2(Y[j], Y_o[j]) < PER_LAYER_TOL, (m, j)


@pytest.mark.parametrize("case", [
    (160, 256, 12, 20, 48.0, 30.0, 10, 20, 4),    # nC = 20: chunked pair histogram in the table Gram
    (256, 256, 30, 30, 40.0, 30.0, 6, 40, 6),     # BASELINE configs[4] sample grid (30 x 30 = 900 samples, 6 weights)
    (192, 320, 20, 20, 60.0, 30.0, 8, 50, 4),     # BASELINE configs[2] sample grid (20 x 20 = 400 samples)
])
def test_large_sample_grids_on_the_table_path(nle, oracle, ctx, case):
    """more than 256 samples (or more than 11 sample columns): only the table formulation is Phi-free there"""
    H, W, nr, nc, hx, hy, T, K, L = case
    x = oracle.synthetic_luminance(H, W)
    V_o, S_o, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
    assert inter["lam"].size == nr * nc and inter["lam"][-1] > 1e-9, "rank cut must not be borderline"
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    for m in (2, 4):
        ctx.set_mode(m)
        ctx.profile(True)
        try:
            f, Y = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
            stats = ctx.kernel_stats()
        finally:
            ctx.profile(False)
            ctx.set_mode(0)
        if m == 2:
            # table builds: one per Sinkhorn pass after the column sum, plus one timed region per k_hist_dot
            # launch of the apply (up to 4 layers per launch, as many tables as fit in 144 KB of LDS)
            nC = nle.sample_grid(H, W, nr, nc)["n_sel_cols"]
            lb = max(1, min(L, 4, (144 * 1024) // (256 * (nC | 1) * 8)))
            assert stats["sink_tables"][0] == 2 * T - 1 + -(-L // lb) and stats["gram_gemm"][0] == 1
        assert f.info()["p"] == nr * nc and f.info()["K"] == S_o.size
        # both fp64 formulations meet the bar here (lambda_min(Ka) ~ 1e-7: the fp32 forms do not)
        assert rel_l2(f.eigvals, S_o) < 1e-5, m
        for j in range(L):
            assert rel_l2(Y[j], Y_o[j]) < PER_LAYER_TOL, (m, j)
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [True, False])
def test_host_buffer_entry_points_match_the_device_ones(nle, oracle, ctx, pinned):
    """nle_train_host + nle_apply_layers_host (SURVEY.md section 8d's host plane -> host layers path): same numbers as
    the device-pointer entry points, with x = NULL (the kept training plane) and with x passed again; per-layer
    downloads run on the copy stream"""
    H, W, nr, nc, hx, hy, T, K, L = 120, 160, 6, 8, 40.0, 30.0, 6, 12, 5
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    f_dev = nle.NLEFilter(ctx).train_filter(x, nr, nc, hx, hy, T, K)
    Y_dev = f_dev.apply_layers(x, L).cpu().numpy()
    if pinned:
        h_x = ctx.host_alloc((H, W))
        h_x[...] = x
        h_y = ctx.host_alloc((L, H * W))
    else:
        h_x, h_y = x.copy(), np.empty((L, H * W), dtype=np.float32)
    f = nle.NLEFilter(ctx).train_filter_host(h_x, nr, nc, hx, hy, T, K)
    np.testing.assert_allclose(f.eigvals, f_dev.eigvals, rtol=1e-9)
    h_y[...] = -1.0
    f.apply_layers_host(None, L, h_y)
    assert rel_l2(h_y, Y_dev) < 1e-6
    h_y[...] = -1.0
    f.apply_layers_host(h_x, L, h_y)
    assert rel_l2(h_y, Y_dev) < 1e-6
    # a filter trained from a device pointer keeps no plane: NULL is refused, loudly
    with pytest.raises(nle.NLEError):
        f_dev.apply_layers_host(None, L, h_y)
    # the one-plane return `enhance` needs (src/filter.cpp:428-436): apply, clamp, convertTo(CV_8U) on the device, n bytes
    # home.  The clamp and the round-half-even act on the fp64 value of the plane (nothing is rounded to fp32 first), so the
    # bytes are the ORACLE's fp64 pipeline's -- not merely the rounding of the fp32 plane, from which they may differ at ties
    w = [2.0, 3.0, 3.0, 4.0, 1.0]
    fs = nle.transform_eigenvalues(f.eigvals, w)
    V_o, S_o = oracle.train_filter(x.astype(np.float64), nr, nc, hx, hy, T, K)
    want = np.rint(np.clip(oracle.apply_filter(V_o, x.astype(np.float64), oracle.transform_eigenvalues(S_o, w)), 0, 255)).astype(np.uint8).ravel()
    assert 0 in want and 255 in want                      # the weights push parts of the plane out of range: the clamp is exercised
    got = f_dev.apply_u8(x, fs).cpu().numpy()
    assert int((got != want).sum()) <= 1 and np.abs(got.astype(int) - want.astype(int)).max() <= 1      # a tie at 1e-9 at most
    y32 = np.rint(np.clip(f_dev.apply(x, fs).cpu().numpy(), 0, 255)).astype(np.uint8)
    assert (got != y32).mean() < 1e-3 and np.abs(got.astype(int) - y32.astype(int)).max() <= 1          # the fp32 plane's ties
    lv = f_dev.apply_rounded8(x, fs).cpu().numpy()                                                      # the same levels as fp32
    assert lv.dtype == np.float32 and np.array_equal(lv, got.astype(np.float32))
    h_o = ctx.host_alloc((H * W,), dtype=np.uint8) if pinned else np.empty(H * W, dtype=np.uint8)
    h_o[...] = 7
    f.apply_u8_host(None, fs, h_o)
    assert np.array_equal(h_o, got)
    h_o[...] = 7
    f.apply_u8_host(h_x, fs, h_o)
    assert np.array_equal(h_o, got)
    with pytest.raises(nle.NLEError):
        f_dev.apply_u8_host(None, fs, h_o)
    with pytest.raises(nle.NLEError):
        f.apply_u8_host(None, fs, np.empty(H * W - 1, dtype=np.uint8))
    # the 8-bit plane itself as input (nle_train_host_u8: what getLuminanceChannel hands over, src/filter.cpp:460-469):
    # the same filter bit for bit, and it keeps the plane for the NULL forms like nle_train_host
    if np.array_equal(x, np.rint(x)) and x.min() >= 0 and x.max() <= 255:
        h_x8 = ctx.host_alloc((H, W), dtype=np.uint8) if pinned else np.empty((H, W), dtype=np.uint8)
        h_x8[...] = x.astype(np.uint8)
        f8 = nle.NLEFilter(ctx).train_filter_host_u8(h_x8, nr, nc, hx, hy, T, K)
        assert np.array_equal(f8.eigvals, f.eigvals) and f8.diag() == f.diag()
        h_o[...] = 7
        f8.apply_u8_host(None, fs, h_o)
        assert np.array_equal(h_o, got)
        h_y[...] = -1.0
        f8.apply_layers_host(None, L, h_y)
        assert rel_l2(h_y, Y_dev) < 1e-6
        f8.close()
    f.close()
    f_dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["noise", "flat", "two-level", "tiny", "wide20", "wide36", "mixed", "mixed4k"])
def test_level_sorted_rows_agree_with_the_histogram_kernels_and_are_bitwise_reproducible(nle, oracle, kind):
    """sorted.hip (level-sorted rows, register accumulation, fixed combine tree) against the LDS-atomic histogram
    kernels it replaces (NLE_NO_SORTED_ROWS=1), on images that stress the chunking: noise, one flat level (every
    thread of a row on one level), a two-level checkerboard, and an image narrower than a workgroup.  The sorted form
    has no atomics, so two runs must agree bit for bit.  "mixed": flat rows (a level of more than 64 chunks: the pass
    kernel's eight waves meet at a barrier) interleaved with noisy and smooth rows (wave-aligned chunks: every wave
    combines its own levels, no barrier, the waves drift apart) -- the two kinds of row alternate inside one workgroup's
    row list; "mixed4k" the same at 4096 columns, two workgroups per CU and eight rows per workgroup."""
    rng = np.random.default_rng(5)
    if kind == "tiny":
        H, W, nr, nc = 40, 37, 4, 5
    elif kind == "wide20":           # more than 12 sample columns: column factors by recurrence, pair tables in
        H, W, nr, nc = 150, 700, 6, 20          # several launches (k_sorted_gram_wide)
    elif kind == "wide36":
        H, W, nr, nc = 150, 700, 4, 36
    else:
        H, W, nr, nc = 150, 700, 6, 10
    if kind == "mixed4k":
        H, W, nr, nc = 2100, 4096, 6, 10
    if kind in ("mixed", "mixed4k"):
        x = rng.integers(0, 256, (H, W)).astype(np.float32)
        x[1::3] = 97.0                                        # flat rows ...
        x[1::3, ::5] = 140.0
        cc = np.arange(W, dtype=np.float32)[None, :]
        x[2::3] = np.clip(np.rint(100.0 + 60.0 * np.sin(cc / 97.0) + rng.integers(-12, 13, x[2::3].shape)), 0, 255)   # ... and smooth + noise ones
    elif kind in ("noise", "tiny", "wide20", "wide36"):
        x = rng.integers(0, 256, (H, W)).astype(np.float32)
    elif kind == "flat":
        x = np.full((H, W), 97.0, dtype=np.float32)
        x[::7, ::5] = 140.0          # a few other pixels so that Ka is not singular
        x += rng.integers(0, 2, (H, W)).astype(np.float32) * (rng.random((H, W)) < 0.01)
    else:
        rr, cc = np.mgrid[0:H, 0:W]
        x = np.where((rr + cc) & 1, 60.0, 200.0).astype(np.float32)
        x[rng.random((H, W)) < 0.02] = 128.0
    hx, hy, T, K, L = W / 3.0, 40.0, 5, 8, 3

    def run():
        c = nle.Context(0)
        f = nle.NLEFilter(c).train_filter(x, nr, nc, hx, hy, T, K)
        ev = f.eigvals.copy()
        Y = f.apply_layers(x, L).cpu().numpy()
        f.close()
        c.close()
        return ev, Y

    ev1, Y1 = run()
    ev2, Y2 = run()
    assert np.array_equal(ev1, ev2) and np.array_equal(Y1, Y2)          # bitwise reproducible
    os.environ["NLE_NO_SORTED_ROWS"] = "1"
    try:
        ev0, Y0 = run()
    finally:
        del os.environ["NLE_NO_SORTED_ROWS"]
    assert rel_l2(ev1, ev0) < 1e-9
    for j in range(L):
        assert rel_l2(Y1[j], Y0[j]) < 1e-6, (kind, j)
    # the pass kernel on moments (two table reads per pixel, the default where no power leaves fp64's normal range) / column
    # factors by recurrence, against all column factors read from the table (NLE_SORTED_TABLE=1): O(nC^2) ulp apart
    os.environ["NLE_SORTED_TABLE"] = "1"
    try:
        ev3, Y3 = run()
    finally:
        del os.environ["NLE_SORTED_TABLE"]
    assert rel_l2(ev1, ev3) < 1e-10, rel_l2(ev1, ev3)   # (1.2e-11 on the ill-conditioned "mixed" image, 1e-13 on noise)
    for j in range(L):
        assert rel_l2(Y1[j], Y3[j]) < 1e-7, (kind, j, rel_l2(Y1[j], Y3[j]))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["non-integer", "wide grid", "many eigenvectors"])
def test_auto_mode_falls_back_to_the_fp64_decomposition(nle, oracle, ctx, kind):
    """whatever the table form cannot take -- a luminance plane that is not integer valued, a sample grid wider than
    36 columns, more than 128 eigenvectors -- auto mode runs in fp64 too (NLE_MODE_MATERIALISED_F64) and meets the
    bar; it never selects an fp32 formulation"""
    rng = np.random.default_rng(11)
    if kind == "non-integer":
        H, W, nr, nc, hx, hy, T, K, L = 60, 80, 5, 6, 6.0, 25.0, 6, 12, 4      # a few pixels of spatial bandwidth
        x = oracle.synthetic_luminance(H, W) + rng.random((H, W)) * 0.75
    elif kind == "wide grid":
        H, W, nr, nc, hx, hy, T, K, L = 40, 200, 3, 40, 30.0, 30.0, 5, 10, 3
        x = oracle.synthetic_luminance(H, W)
    else:
        H, W, nr, nc, hx, hy, T, K, L = 64, 64, 12, 12, 12.0, 30.0, 5, 140, 3
        x = oracle.synthetic_luminance(H, W)
    x = x.astype(np.float32).astype(np.float64)      # the ABI takes fp32 planes: compare on the same values
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f, Y = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    d = f.diag()
    assert d["formulation"] == nle.MODE_MATERIALISED_F64 and d["K"] == S_o.size
    assert rel_l2(f.eigvals, S_o) < 1e-9
    errs = [rel_l2(Y[j], Y_o[j]) for j in range(L)]
    print(kind, "per-layer", ["%.1e" % e for e in errs])
    assert max(errs) < 1e-6          # fp64 throughout: only the fp32 output planes round
    V = f.eigvecs().cpu().numpy()[:, :S_o.size].astype(np.float64)
    assert rel_l2(_align_signs(V, V_o), V_o) < 1e-6
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(96, 128, 6, 8, 32.0, 30.0, 10, 10, 4), (267, 400, 10, 20, 100.0, 30.0, 12, 30, 4),
                                  (120, 300, 5, 33, 80.0, 25.0, 6, 20, 3), (64, 64, 3, 1, 20.0, 30.0, 5, 3, 2)])
def test_gram_by_index_sums_equals_the_pair_table_form(nle, oracle, ctx, case):
    """On the equispaced sample grid the product of two column (row) factors depends on the pair only through the sum of
    its indices (sorted.hip: k_sorted_gsum): the Gram stage then needs 2 nC - 1 tables and 2 nR - 1 GEMM rows instead of
    nC (nC + 1) / 2 and nR (nR + 1) / 2.  Exact algebra: against the pair-table form (NLE_GRAM_PAIRS=1) eigenvalues and
    layers agree to rounding, and both meet the oracle."""
    H, W, nr, nc, hx, hy, T, K, L = case
    x = oracle.synthetic_luminance(H, W)
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    ctx.set_mode(2)
    try:
        f1, Y1 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
        os.environ["NLE_GRAM_PAIRS"] = "1"
        try:
            f2, Y2 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
        finally:
            del os.environ["NLE_GRAM_PAIRS"]
    finally:
        ctx.set_mode(0)
    assert f1.diag() == f2.diag()
    assert rel_l2(f1.eigvals, f2.eigvals) < 1e-10
    for j in range(L):
        assert rel_l2(Y1[j], Y2[j]) < 1e-6, (j, rel_l2(Y1[j], Y2[j]))
        assert rel_l2(Y1[j], Y_o[j]) < 1e-4
    f1.close()
    f2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("where", [(0, 0), (0, 1), (17, 63), (40, 129), (95, 159), (33, 2)])
def test_one_non_integer_pixel_anywhere_keeps_the_plane_off_the_tables(nle, oracle, ctx, where):
    """auto mode takes the table formulation only when EVERY pixel is an integer in [0, 255] (k_check_levels): one pixel
    that is not -- in any lane of any wave, at either end of an unaligned plane -- sends the plane to the general fp64 form,
    and the result is still the oracle's"""
    H, W, nr, nc, hx, hy, T, K, L = 96, 160, 6, 8, 40.0, 30.0, 6, 8, 3
    x = oracle.synthetic_luminance(H, W).copy()
    x[where] += 0.5
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f, Y = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    assert f.diag()["formulation"] in (nle.MODE_MATERIALISED_F64, nle.MODE_STREAMED_F64)
    for j in range(L):
        assert rel_l2(Y[j], Y_o[j]) < 1e-4
    f.close()
    x2 = oracle.synthetic_luminance(H, W).copy()
    x2[where] = 256.0      # an integer, but not a level
    f2, _ = _run_device(nle, ctx, x2, nr, nc, hx, hy, T, K, L)
    assert f2.diag()["formulation"] in (nle.MODE_MATERIALISED_F64, nle.MODE_STREAMED_F64)
    f2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("levels", [(96, 111), (120, 135), (0, 255), (3, 40), (250, 255), (77, 77)],
                         ids=["one tile", "across two tiles", "full range", "low", "top tile", "flat"])
def test_level_tiles_that_do_not_occur_are_skipped_exactly(nle, oracle, ctx, levels):
    """The g / h tables of the Sinkhorn and apply passes have a column per (sample column, level); the columns of 16-level
    tiles that occur nowhere in the image are neither made, stored nor contracted (check_levels reports the tiles).  Same
    result as with all 16 tiles (NLE_ALL_LEVEL_TILES=1), to rounding, whatever the range -- and the oracle's."""
    H, W, nr, nc, hx, hy, T, K, L = 96, 160, 6, 8, 40.0, 12.0, 8, 10, 3
    lo, hi = levels
    if hi - lo > 100:
        hy = 40.0   # a photometric bandwidth in proportion to the range (12 on 0..255 is a numerically wild example)
    base = oracle.synthetic_luminance(H, W)
    if hi > lo:
        x = np.rint(lo + (base - base.min()) * ((hi - lo) / max(base.max() - base.min(), 1.0)))
        x[0, 0], x[-1, -1] = lo, hi
    else:
        x = np.full((H, W), float(lo))
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f1, Y1 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    os.environ["NLE_ALL_LEVEL_TILES"] = "1"
    try:
        f2, Y2 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    finally:
        del os.environ["NLE_ALL_LEVEL_TILES"]
    assert f1.diag() == f2.diag() and f1.diag()["formulation"] == nle.MODE_PHI_FREE
    assert f1.eigvals.size == S_o.size
    assert rel_l2(f1.eigvals, f2.eigvals) < 1e-11
    assert rel_l2(f1.eigvals, S_o) < 1e-8
    for j in range(L):
        assert rel_l2(Y1[j], Y2[j]) < 1e-7, (j, rel_l2(Y1[j], Y2[j]))
        if np.linalg.norm(Y_o[j]) > 1e-9:
            assert rel_l2(Y1[j], Y_o[j]) < 1e-4, (j, rel_l2(Y1[j], Y_o[j]))
    f1.close()
    f2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["non-integer", "wide grid", "many eigenvectors"])
def test_streamed_fp64_form_takes_what_does_not_fit(nle, oracle, ctx, kind):
    """NLE_MODE_STREAMED_F64: the fp64 fallback WITHOUT the N x r matrix (affinity rows regenerated chunk by chunk, bounded
    workspace) -- what auto mode takes when Phi would not fit the device instead of refusing.  Same three kinds of input
    as above, forced into it through NLE_AUTO_STREAM64 with a chunk budget of 1 MB so that a pass walks several chunks;
    it must meet the same bar and agree with the materialised fp64 form."""
    rng = np.random.default_rng(11)
    if kind == "non-integer":
        H, W, nr, nc, hx, hy, T, K, L = 60, 80, 5, 6, 6.0, 25.0, 6, 12, 4
        x = oracle.synthetic_luminance(H, W) + rng.random((H, W)) * 0.75
    elif kind == "wide grid":
        H, W, nr, nc, hx, hy, T, K, L = 40, 200, 3, 40, 30.0, 30.0, 5, 10, 3
        x = oracle.synthetic_luminance(H, W)
    else:
        H, W, nr, nc, hx, hy, T, K, L = 64, 64, 12, 12, 12.0, 30.0, 5, 140, 3
        x = oracle.synthetic_luminance(H, W)
    x = x.astype(np.float32).astype(np.float64)
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f4, Y4 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    assert f4.diag()["formulation"] == nle.MODE_MATERIALISED_F64
    os.environ["NLE_AUTO_STREAM64"] = "1"
    os.environ["NLE_STREAM64_CHUNK_MB"] = "1"
    try:
        f, Y = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    finally:
        del os.environ["NLE_AUTO_STREAM64"], os.environ["NLE_STREAM64_CHUNK_MB"]
    d = f.diag()
    assert d["formulation"] == nle.MODE_STREAMED_F64 and d["K"] == S_o.size
    assert d["r_Ka"] == f4.diag()["r_Ka"] and d["r_Wa"] == f4.diag()["r_Wa"] and d["r_Q"] == f4.diag()["r_Q"]
    assert rel_l2(f.eigvals, S_o) < 1e-9
    errs = [rel_l2(Y[j], Y_o[j]) for j in range(L)]
    print(kind, "streamed per-layer", ["%.1e" % e for e in errs])
    assert max(errs) < 1e-6
    for j in range(L):
        assert rel_l2(Y[j], Y4[j]) < 1e-6
    f.close()
    f4.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [SMALL_CASES[1], SMALL_CASES[4], (96, 128, 8, 10, 40.0, 30.0, 8, 79, 4),
                                  (96, 128, 6, 8, 32.0, 30.0, 60, 10, 4)])
def test_lanczos_topk_solver_for_Q_matches_the_oracle_s_use_spectra_branch(nle, oracle, ctx, case):
    """the reference's USE_SPECTRA build finds the top eigenpairs of Q by Lanczos
    (src/filter.cpp:170-199, 310-311).  Opt-in here (nle_ctx_set_topk_solver / NLE_Q_SOLVER=lanczos) and held to the ORACLE's
    restatement of that branch (oracle.train_filter(use_spectra=True): nev = min(K, q - 1), ncv = min(2 nev, q), converged
    pairs only, cut at 1e-10, products with the unsymmetrised Q): the same K', eigenvalues to 1e-8 + 0.25 ||Wa - Wa^T||_F,
    every layer to the 1e-4 bar + 2 ||Wa - Wa^T||_F.  The second terms are what a SYMMETRIC Lanczos iteration leaves undefined on a matrix that is
    not symmetric: Q = Wa + S (Wab Wab^T) S inherits the asymmetry of Wa (T = 10 Sinkhorn iterations leave 1e-5; T = 60,
    the last case, 5e-12), the tridiagonal projection drops entries of that size, and which ones depends on the start
    vector -- the oracle's two branches themselves differ by 0.06 ||Wa - Wa^T||_F (the full solver reads the lower
    triangle, :207).  Beside it the default full solver, where the two builds must agree: K' equal unless K >= q."""
    H, W, nr, nc, hx, hy, T, K, L = case
    x = oracle.synthetic_luminance(H, W)
    _, S_d, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
    asym = float(np.linalg.norm(inter["Wa"] - inter["Wa"].T))
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K, use_spectra=True)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    f0, Y0 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    ctx.set_topk_solver(1)
    try:
        f1, Y1 = _run_device(nle, ctx, x, nr, nc, hx, hy, T, K, L)
    finally:
        ctx.set_topk_solver(0)
    q = f0.diag()["r_Ka"]
    k0, k1 = f0.info()["K"], f1.info()["K"]
    assert k1 == S_o.size == min(k0, q - 1)
    tol = 1e-8 + 0.25 * asym
    print(f"||Wa - Wa^T||_F = {asym:.2e}; Lanczos option vs oracle's USE_SPECTRA branch: max |d lambda| = "
          f"{np.abs(f1.eigvals - S_o).max():.2e} (oracle's two branches: {np.abs(S_o - S_d[:k1]).max():.2e})")
    assert np.abs(f1.eigvals - S_o).max() < tol
    for j in range(L):
        assert rel_l2(Y1[j], Y_o[j]) < PER_LAYER_TOL + 2.0 * asym, j
    assert np.abs(f1.eigvals - f0.eigvals[:k1]).max() < tol
    if T >= 50:   # Sinkhorn converged: Wa is symmetric to rounding and every solver must give the same filter
        assert np.abs(f1.eigvals - S_o).max() < 1e-8
        for j in range(L):
            assert rel_l2(Y1[j], Y0[j]) < 1e-6, j
    f0.close()
    f1.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(16, 4), (57, 20), (196, 50), (200, 100), (224, 7)])
def test_device_tridiagonalisation_gives_the_host_solver_s_eigenpairs(nle, ctx, n, k):
    """k_tridiag (one workgroup, the matrix in registers) + the host's QL / inverse iteration / back-transformation against
    numpy and against the all-host solver: eigenvalues of a matrix with a spectrum decaying to 1e-12 like Q's, a leading
    block of eigenvectors, only the lower triangle read; also a matrix with zero rows (the scale == 0 branch)."""
    rng = np.random.default_rng(n * 31 + k)
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.sort(np.concatenate([1.0 - 0.5 * rng.random(n // 3), 10.0 ** (-12 * rng.random(n - n // 3))]))[::-1]
    M = (Qo * lam) @ Qo.T
    M = 0.5 * (M + M.T)
    Mlow = np.tril(M) + 7.0 * np.triu(rng.standard_normal((n, n)), 1)      # garbage above the diagonal must be ignored
    U, D, r = ctx.eigen_decomposition_top_device(Mlow, k)
    Uh, Dh, rh = nle.eigen_decomposition_top(Mlow, k)
    w = np.linalg.eigvalsh(M)[::-1]
    assert np.abs(D - w).max() < 1e-13 and np.abs(D - Dh).max() < 1e-13 and r == rh
    assert np.abs(U.T @ U - np.eye(k)).max() < 1e-10
    assert np.abs(M @ U - U * D[:k]).max() < 1e-12
    # decoupled matrix: a zero row / column in the middle and a diagonal tail
    M2 = M.copy()
    M2[n // 2, :] = 0.0
    M2[:, n // 2] = 0.0
    M2[: n // 4, n // 4:] = 0.0
    M2[n // 4:, : n // 4] = 0.0
    U2, D2, _ = ctx.eigen_decomposition_top_device(M2, k)
    assert np.abs(D2 - np.linalg.eigvalsh(M2)[::-1]).max() < 1e-13
    assert np.abs(M2 @ U2 - U2 * D2[:k]).max() < 1e-12
