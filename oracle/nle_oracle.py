"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A literal fp64 numpy restatement of the nonlocal-filter hot path of
lightalchemist/nonlocal-image-edit (reference tree `/root/reference`, cited as
`src/filter.cpp:LINE`).  It exists to CHECK the MI355X HIP path, never to be it:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product (`nonlocal-image-edit_amd/`) never does.

Pinning (see DESIGN.md "Oracle"):
  * The reference cannot be compiled here (needs Eigen3 + OpenCV, neither in the
    image -- an ordinary missing dependency), so there is no
    `oracle/_ref`.
  * This restatement is pinned by the reference's own unit tests
    (`test/test_filter.cpp`: 3x3 eigen KAT, Sinkhorn row/col-sum properties,
    orthogonalize V^T V = I, conversion order) -- `tests/test_oracle_reference_cases.py`
    -- and end-to-end by ALL ELEVEN sample pairs of the reference's README (input image,
    arguments, the author's output image; `tests/test_oracle_readme_pairs.py`): the oracle's
    output images are within 0.009 .. 0.044 grey levels (mean |dL|, p99 <= 1) of the author's,
    and byte for byte identical to them up to rounding ties of the filtered L plane (round 4).
  * `computeKernel`, `nystromApproximation`, `transformEigenValues`, `apply` have no
    reference unit test: for those the README pairs are the pin.

Third-party arithmetic restated: Eigen `SelfAdjointEigenSolver` (lower triangle,
ascending) -> `numpy.linalg.eigh(UPLO="L")`; Eigen dense products -> numpy matmul;
OpenCV's 8-bit BGR -> Lab (`cv::cvtColor`, imgproc `RGB2Lab_b`: a fixed-point table
algorithm, not the float formula of the documentation) -> `bgr_to_lab8`, which is what
brought the README pairs from 0.2 .. 3.4 grey levels to the figures above; OpenCV's 8-bit
Lab -> BGR (imgproc `Lab2RGBinteger`, integer tables too) -> `lab8_to_bgr`, which reproduces the author's output FILES
byte for byte where the filtered L plane has no rounding tie (flower, brickwall: every byte).
Everything is fp64 like the reference (`include/filter.hpp:10-14`).

Two forms are provided:
  * the LITERAL form: the reference's `[selected; rest]` row order and its N x p
    temporaries (`compute_kernel` ... `train_filter`), usable up to ~1 MP;
  * the STREAMING form (`train_filter_streaming`): same math in natural pixel order,
    tiled over pixels, never holding N x p -- this is the CPU baseline for 2048^2+ and
    mirrors the decomposition the GPU path uses (and, with `shard=(g, G)`, the
    per-rank slab decomposition of the multi-GPU path).
"""
from __future__ import annotations

import numpy as np

EPS = 1e-10  # include/filter.hpp:14


# --------------------------------------------------------------------------- helpers
def inplace_reciprocal(v: np.ndarray, eps: float = EPS):
    """src/filter.cpp:42-54 -- 1/v where |v| >= eps, else 0; returns (out, nnz)."""
    v = np.asarray(v, dtype=np.float64)
    mask = np.abs(v) >= eps
    out = np.zeros_like(v)
    out[mask] = 1.0 / v[mask]
    return out, int(mask.sum())


def sample_grid(nrows: int, ncols: int, n_row_samples: int, n_col_samples: int):
    """Closed form of the predicate in `samplePixels`, src/filter.cpp:56-71.

    Returns (sel_rows, sel_cols): the selected pixel set is their Cartesian product.
    C++ int division truncates toward zero; all operands here are non-negative.
    """
    row_step = nrows // n_row_samples
    col_step = ncols // n_col_samples
    row_off = (row_step - 1 + (nrows - row_step * n_row_samples)) // 2
    col_off = (col_step - 1 + (ncols - col_step * n_col_samples)) // 2
    r = np.arange(nrows)
    c = np.arange(ncols)
    sel_r = r[(r >= row_off) & (r <= nrows - row_off) & ((r - row_off) % row_step == 0)]
    sel_c = c[(c >= col_off) & (c <= ncols - col_off) & ((c - col_off) % col_step == 0)]
    return sel_r, sel_c


def sample_pixels(nrows: int, ncols: int, n_row_samples: int, n_col_samples: int):
    """src/filter.cpp:56-80 -- (selected, rest) as flat row-major pixel indices, both in
    row-major scan order."""
    sel_r, sel_c = sample_grid(nrows, ncols, n_row_samples, n_col_samples)
    mask = np.zeros((nrows, ncols), dtype=bool)
    mask[np.ix_(sel_r, sel_c)] = True
    flat = mask.ravel()
    idx = np.arange(nrows * ncols)
    return idx[flat], idx[~flat]


def _neg_weighted_distance(img, rows_a, cols_a, vals_a, rows_b, cols_b, vals_b, sw, pw):
    """src/filter.cpp:104-112, vectorised: spatial term in integers then promoted."""
    dr = rows_a[:, None].astype(np.int64) - rows_b[None, :].astype(np.int64)
    dc = cols_a[:, None].astype(np.int64) - cols_b[None, :].astype(np.int64)
    sq_sp = (dr * dr + dc * dc).astype(np.float64)
    di = vals_a[:, None] - vals_b[None, :]
    return -sw * sq_sp - pw * (di * di)


# --------------------------------------------------------------------- literal stages
def compute_kernel(mat: np.ndarray, n_row_samples: int, n_col_samples: int,
                   hx: float, hy: float, chunk: int = 1 << 16):
    """`computeKernel`, src/filter.cpp:114-167.

    mat: H x W fp64 luminance.  Returns (perm, Ka, Kab): perm[i] = row-major index of
    the i-th pixel in [selected; rest] order (`P.indices()`, :156-164); Ka p x p;
    Kab p x (N-p).
    """
    mat = np.asarray(mat, dtype=np.float64)
    H, W = mat.shape
    if n_row_samples > H or n_col_samples > W:  # :117-119
        raise RuntimeError("Number of samples per row and col must be <= that of image.")
    sel, rest = sample_pixels(H, W, n_row_samples, n_col_samples)
    pw = 1.0 / (hy * hy)  # :128
    sw = 1.0 / (hx * hx)  # :129
    flat = mat.ravel()
    sr, sc, sv = sel // W, sel % W, flat[sel]
    Ka = np.exp(_neg_weighted_distance(mat, sr, sc, sv, sr, sc, sv, sw, pw))  # :133-137,144
    Kab = np.empty((sel.size, rest.size), dtype=np.float64)
    for s in range(0, rest.size, chunk):  # :139-141,145
        rr = rest[s:s + chunk]
        Kab[:, s:s + chunk] = np.exp(
            _neg_weighted_distance(mat, sr, sc, sv, rr // W, rr % W, flat[rr], sw, pw))
    perm = np.concatenate([sel, rest])
    return perm, Ka, Kab


def eigen_decomposition(M: np.ndarray, eps: float = EPS, info: list | None = None, force_rank: int | None = None):
    """`eigenDecomposition`, src/filter.cpp:204-228: lower-triangle symmetric eigensolve,
    descending order, keep the leading run with D >= eps.  Returns (U, D).

    Diagnostics only (never used by the parity path): `info`, when given, receives a dict with the size, the
    number kept and the eigenvalues either side of the cut; `force_rank` overrides the cut (used by the README-pair
    study to ask which rank the author's Eigen build must have kept)."""
    M = np.asarray(M, dtype=np.float64)
    w, v = np.linalg.eigh(M, UPLO="L")
    D = w[::-1]
    U = v[:, ::-1]
    r = 0
    while r < D.size and D[r] >= eps:  # :214
        r += 1
    if info is not None:
        info.append(dict(n=int(D.size), kept=r, last_kept=float(D[r - 1]) if r else None,
                         first_dropped=float(D[r]) if r < D.size else None,
                         near_cut=[float(v) for v in D if 0.9 * eps <= v <= 1.1 * eps]))
    if force_rank is not None:
        r = int(force_rank)
    return np.ascontiguousarray(U[:, :r]), D[:r].copy()


def nystrom_approximation(Ka: np.ndarray, Kab: np.ndarray, info: list | None = None, force_rank: int | None = None):
    """`nystromApproximation`, src/filter.cpp:257-280.  Returns (eigvals (r), phi (N x r)),
    phi = [V ; Kab^T V diag(1/lambda)] in [selected; rest] order."""
    eigvecs, eigvals = eigen_decomposition(Ka, info=info, force_rank=force_rank)
    inv, nnz = inplace_reciprocal(eigvals)  # :265-266
    eigvecs = eigvecs[:, :nnz]  # :269
    eigvals = eigvals[:nnz]  # :271
    phi = np.vstack([eigvecs, (Kab.T @ eigvecs) * inv[:nnz][None, :]])  # :275
    return eigvals, phi


def sinkhorn(phi: np.ndarray, eigvals: np.ndarray, max_iter: int = 10):
    """`sinkhorn`, src/filter.cpp:230-254.  Returns (Wa q x q, Wab q x (N-q)), q = phi.cols()."""
    Wa, Wab, _, _ = sinkhorn_with_scalings(phi, eigvals, max_iter)
    return Wa, Wab


def sinkhorn_with_scalings(phi, eigvals, max_iter=10):
    phi = np.asarray(phi, dtype=np.float64)
    eigvals = np.asarray(eigvals, dtype=np.float64)
    n = phi.shape[0]
    r = np.ones(n)
    c = np.zeros(n)
    for _ in range(max_iter):  # :238-245
        c, _ = inplace_reciprocal(phi @ (eigvals * (phi.T @ r)))
        r, _ = inplace_reciprocal(phi @ (eigvals * (phi.T @ c)))
    p = phi.shape[1]  # :247  (q = number of retained eigenpairs, NOT the sample count)
    left = (r[:p, None] * (phi[:p] * eigvals[None, :]))  # R * (phi_top * D)
    Wa = left @ (c[:p, None] * phi[:p]).T  # :249
    Wab = left @ (c[p:, None] * phi[p:]).T  # :250
    return Wa, Wab, r, c


def _spectra_start_vector(n: int) -> np.ndarray:
    """The start vector of Spectra's `init()` (ext/Spectra/SymEigsBase.h:297-302: `SimpleRandom<Scalar> rng(0)`): the
    minimal-standard Lehmer generator x <- 16807 x mod (2^31 - 1) from x = 1 (a zero seed is replaced by 1), each value
    x / (2^31 - 1) - 0.5.  Only the path of the iteration depends on it, not its limit."""
    m = (1 << 31) - 1
    x, out = 1, np.empty(n)
    for i in range(n):
        x = (16807 * x) % m
        out[i] = x / m - 0.5
    return out


def topk_eigen_decomposition(M: np.ndarray, n_largest: int, eps: float = EPS, tol: float = 1e-10, maxit: int = 1000,
                             info: list | None = None):
    """`topkEigenDecomposition`, src/filter.cpp:170-199 -- what `orthogonalize` calls on Q in a USE_SPECTRA build (:310-311):
    `Spectra::SymEigsSolver<double, LARGEST_MAGN, DenseGenMatProd>` with nev = min(nLargest, n - 1) (:171) and
    ncv = min(2 nev, n) (:173), `compute()` with its defaults (at most 1000 restarts, tolerance 1e-10, result sorted by
    algebraic value descending: ext/Spectra/SymEigsBase.h:331), then only the CONVERGED pairs (:379-398), then the leading
    run >= eps (:187-196).  Returns (U (n x r), D (r)).

    Spectra (vendored in the reference, but it needs Eigen to compile) is an implicitly restarted Lanczos method.  Restated
    here as its published algorithm in the mathematically equivalent thick-restart form: an ncv-step Lanczos factorisation
    M V = V T + f e^T with full re-orthogonalisation, driven ONLY by products M x with the matrix AS GIVEN
    (`DenseGenMatProd`: no triangle is mirrored, unlike `eigenDecomposition`, :207); Ritz pairs of T ordered by magnitude;
    a wanted pair counts as converged when |last component of its Ritz vector| ||f|| < tol max(eps^(2/3), |theta|)
    (:111-119); restart on the leading nev' Ritz vectors, nev' adjusted as ARPACK's dsaup2 does (:122-141)."""
    M = np.asarray(M, dtype=np.float64)
    n = M.shape[0]
    nev = min(int(n_largest), n - 1)                    # :171
    if nev < 1:
        raise RuntimeError("topkEigenDecomposition needs a matrix of order >= 2")   # the reference asserts (:172)
    ncv = min(2 * nev, n)                               # :173
    if ncv <= nev:
        raise ValueError("ncv must satisfy nev < ncv <= n")   # Spectra's constructor throws (SymEigsBase.h:270-271)
    eps23 = np.finfo(np.float64).eps ** (2.0 / 3.0)
    near0 = np.finfo(np.float64).tiny * 10.0
    V = np.zeros((n, ncv))
    T = np.zeros((ncv, ncv))
    v0 = _spectra_start_vector(n)
    V[:, 0] = v0 / np.linalg.norm(v0)

    def extend(k0, f):
        """Lanczos steps k0 .. ncv-1 (column k0 of V is set); returns the residual f of the last step"""
        for j in range(k0, ncv):
            w = M @ V[:, j]
            h = V[:, :j + 1].T @ w
            w = w - V[:, :j + 1] @ h
            for _ in range(5):                          # iterated Gram-Schmidt, as the reference's factorisation does
                c = V[:, :j + 1].T @ w
                if np.abs(c).max() <= np.finfo(np.float64).eps * np.linalg.norm(w):
                    break
                w = w - V[:, :j + 1] @ c
                h = h + c
            T[j, j] = h[j]
            beta = np.linalg.norm(w)
            if j + 1 < ncv:
                if beta < near0:                        # invariant subspace: continue with a vector orthogonal to V
                    rng = np.random.default_rng(j)
                    w = rng.standard_normal(n)
                    w -= V[:, :j + 1] @ (V[:, :j + 1].T @ w)
                    V[:, j + 1] = w / np.linalg.norm(w)
                    beta = 0.0
                else:
                    V[:, j + 1] = w / beta
                T[j + 1, j] = T[j, j + 1] = beta
            f = w
        return f

    f = extend(0, None)
    nconv, it = 0, 0
    theta = S = conv = None
    for it in range(maxit + 1):
        theta, S = np.linalg.eigh(T)
        order = np.argsort(-np.abs(theta), kind="stable")       # LARGEST_MAGN selection
        theta, S = theta[order], S[:, order]
        fnorm = np.linalg.norm(f)
        est = np.abs(S[-1, :]) * fnorm
        conv = est[:nev] < tol * np.maximum(eps23, np.abs(theta[:nev]))
        nconv = int(conv.sum())
        if nconv >= nev or it == maxit:
            break
        k = nev + int(np.sum(est[nev:] / max(fnorm, near0) < near0))   # nev_adjusted, SymEigsBase.h:122-141
        k += min(nconv, (ncv - k) // 2)
        if k == 1 and ncv >= 6:
            k = ncv // 2
        elif k == 1 and ncv > 2:
            k = 2
        k = min(k, ncv - 1)
        # thick restart on the leading k Ritz vectors: T = diag(theta_k) bordered by ||f|| S[-1, :k], next vector f / ||f||
        V[:, :k] = V @ S[:, :k]
        T[:, :] = 0.0
        T[np.arange(k), np.arange(k)] = theta[:k]
        if fnorm < near0:
            break
        V[:, k] = f / fnorm
        T[k, :k] = T[:k, k] = fnorm * S[-1, :k]
        f = extend(k, f)
    lam, vec, ok = theta[:nev], V @ S[:, :nev], conv
    if info is not None:
        info.append(dict(n=n, nev=nev, ncv=ncv, restarts=it, converged=nconv))
    lam, vec = lam[ok], vec[:, ok]                      # converged pairs only (SymEigsBase.h:379-398)
    o = np.argsort(-lam, kind="stable")                 # compute()'s default sort: LARGEST_ALGE (:331)
    lam, vec = lam[o], vec[:, o]
    r = 0
    while r < lam.size and lam[r] >= eps:               # :187-189
        r += 1
    return np.ascontiguousarray(vec[:, :r]), lam[:r].copy()


def orthogonalize(Wa: np.ndarray, Wab: np.ndarray, n_eig_vectors: int = 5, eps: float = EPS, info: list | None = None,
                  use_spectra: bool = False):
    """`orthogonalize`, src/filter.cpp:282-331; default (non-Spectra) branch :313-316, or with `use_spectra` the
    USE_SPECTRA build's :310-311.  Returns (V (N x K'), Sq (K'))."""
    eigvecs, eigvals = eigen_decomposition(Wa, info=info)  # :287
    inv_root, _ = inplace_reciprocal(eigvals)  # :289-291
    inv_root = np.sqrt(inv_root)
    inv_root_wa = (eigvecs * inv_root[None, :]) @ eigvecs.T  # :292
    Q = Wa + inv_root_wa @ (Wab @ Wab.T) @ inv_root_wa  # :296
    if use_spectra:
        Vq, Sq = topk_eigen_decomposition(Q, n_eig_vectors, info=info)  # :311
    else:
        Vq, Sq = eigen_decomposition(Q, info=info)  # :313
        k = min(n_eig_vectors, Vq.shape[1])  # :314
        Vq, Sq = Vq[:, :k], Sq[:k]
    inv_root_sq, _ = inplace_reciprocal(Sq)  # :319-321
    inv_root_sq = np.sqrt(inv_root_sq)
    tmp = np.vstack([Wa, Wab.T])  # :324-325
    V = ((tmp @ inv_root_wa) @ Vq) * inv_root_sq[None, :]  # :327 (left-associative)
    return V, Sq


def transform_eigenvalues(eigvals: np.ndarray, weights) -> np.ndarray:
    """`transformEigenValues`, src/filter.cpp:334-347."""
    eigvals = np.asarray(eigvals, dtype=np.float64)
    fS = np.full(eigvals.shape, float(weights[0]))
    for k in range(1, len(weights)):
        fS = fS + (weights[k] - weights[k - 1]) * np.power(eigvals, float(k))
    return fS


def layer_responses(eigvals: np.ndarray, n_layers: int) -> np.ndarray:
    """Spectral response of each layer implied by `transformEigenValues` (:334-347):
    detail layer j <-> lambda^j - lambda^(j+1), base layer <-> lambda^(L-1).
    sum_j w_j * response_j == transform_eigenvalues(eigvals, w).  Shape (L, K)."""
    eigvals = np.asarray(eigvals, dtype=np.float64)
    out = np.empty((n_layers, eigvals.size))
    for j in range(n_layers - 1):
        out[j] = np.power(eigvals, float(j)) - np.power(eigvals, float(j + 1))
    out[n_layers - 1] = np.power(eigvals, float(n_layers - 1))
    return out


def train_filter(channel: np.ndarray, n_row_samples: int, n_col_samples: int,
                 hx: float, hy: float, n_sinkhorn_iter: int, n_eigen_vectors: int,
                 return_intermediates: bool = False, info: list | None = None, force_rank: int | None = None,
                 use_spectra: bool = False):
    """`NLEFilter::trainFilter`, src/filter.cpp:480-502 (GUI loop :504-511 omitted).
    Returns (eigvecs N x K' in PIXEL order, eigvals K').  `info` collects the three eigensolves' cut
    diagnostics (K_A, W_A, Q, in that order); `force_rank` overrides K_A's cut (diagnostics only)."""
    perm, Ka, Kab = compute_kernel(channel, n_row_samples, n_col_samples, hx, hy)
    eigvals, phi = nystrom_approximation(Ka, Kab, info=info, force_rank=force_rank)
    del Kab
    Wa, Wab, r_vec, c_vec = sinkhorn_with_scalings(phi, eigvals, n_sinkhorn_iter)
    V, S = orthogonalize(Wa, Wab, n_eigen_vectors, info=info, use_spectra=use_spectra)   # use_spectra: the USE_SPECTRA build
    out = np.empty_like(V)
    out[perm] = V  # :502  (P*V).row(P.indices[i]) = V.row(i)
    if return_intermediates:
        return out, S, dict(perm=perm, Ka=Ka, lam=eigvals, phi=phi, Wa=Wa, r=r_vec, c=c_vec)
    return out, S


def apply_filter(eigvecs: np.ndarray, channel: np.ndarray, f_s: np.ndarray) -> np.ndarray:
    """`NLEFilter::apply`, src/filter.cpp:445-458: y = V (diag(fS) V^T x)."""
    if channel.size != eigvecs.shape[0]:  # :447-449
        raise RuntimeError("Number of values in channel must match that of training image.")
    x = np.asarray(channel, dtype=np.float64).ravel()  # row-major flatten, utils.hpp:28-41
    return (eigvecs @ (f_s * (eigvecs.T @ x))).reshape(channel.shape)


def apply_layers(eigvecs, eigvals, channel, n_layers: int) -> np.ndarray:
    """Per-layer outputs y_j = V (resp_j o (V^T x)); shape (L, H, W).  The reference only
    forms sum_j w_j y_j (:428-431); the 1e-4 per-detail-layer bar compares these."""
    x = np.asarray(channel, dtype=np.float64).ravel()
    t = eigvecs.T @ x
    resp = layer_responses(eigvals, n_layers)
    return np.stack([(eigvecs @ (resp[j] * t)).reshape(channel.shape) for j in range(n_layers)])


# ------------------------------------------------------------------ streaming form
def _affinity_rows(lum_flat, W, idx, sr, sc, sv, sw, pw):
    """k_i = exp(negDist(pixel i, sample s)) for a block of pixels: (len(idx), p)."""
    rr, cc = idx // W, idx % W
    dr = rr[:, None] - sr[None, :]
    dc = cc[:, None] - sc[None, :]
    sq = (dr * dr + dc * dc).astype(np.float64)
    di = lum_flat[idx][:, None] - sv[None, :]
    return np.exp(-sw * sq - pw * di * di)


def train_filter_streaming(channel, n_row_samples, n_col_samples, hx, hy,
                           n_sinkhorn_iter, n_eigen_vectors, tile: int = 1 << 15,
                           shard=(0, 1), allreduce=None, phi_dtype=np.float64, intermediates: dict | None = None):
    """Same math as `train_filter` in NATURAL pixel order, tiled over pixels.

    Follows src/filter.cpp:480-502 stage by stage; algebra used to avoid N x p / N x q
    temporaries (all exact identities):
      * phi_i = k_i^T V_A diag(1/lambda) for every non-sample pixel (:275); sample pixels
        take their exact V_A row (:275, top block);
      * Sinkhorn (:238-245) as t = Phi^T y, u = lambda o t, y_i = recip(phi_i . u);
      * Wab Wab^T (:296) = M^T G M with G = sum_{i in B} c_i^2 phi_i phi_i^T,
        M = diag(lambda) Phi_A^T diag(r_A);
      * V_B (:327) = diag(c_B) Phi_B (M S Vq Sq^-1/2); V_A = Wa (S Vq Sq^-1/2).
    The "A block" is the first q = r permuted rows = the first q samples (:247).

    shard=(g, G): only pixel rows [g*H/G, (g+1)*H/G) are processed; `allreduce(x)` must
    then return the sum of x over all G shards (fp64).  Returns (V_local, S) where
    V_local covers this shard's pixels (natural order).  `intermediates`, when given, receives
    lam (eigenvalues of K_A kept), c (this shard's Sinkhorn column scalings, natural order), the
    eigenvalues of W_A and of Q either side of the 1e-10 cut (fixture generation only).
    """
    channel = np.asarray(channel, dtype=np.float64)
    H, W = channel.shape
    if n_row_samples > H or n_col_samples > W:
        raise RuntimeError("Number of samples per row and col must be <= that of image.")
    if allreduce is None:
        allreduce = lambda x: x
    g, G = shard
    row0, row1 = slab_rows(H, g, G)
    lo, hi = row0 * W, row1 * W
    flat = channel.ravel()
    sel_r, sel_c = sample_grid(H, W, n_row_samples, n_col_samples)
    sr = np.repeat(sel_r, sel_c.size)
    sc = np.tile(sel_c, sel_r.size)
    sel = sr * W + sc  # row-major order == permuted order of the samples
    sv = flat[sel]
    sw, pw = 1.0 / (hx * hx), 1.0 / (hy * hy)
    Ka = np.exp(_neg_weighted_distance(channel, sr, sc, sv, sr, sc, sv, sw, pw))
    VA, lam = eigen_decomposition(Ka)
    inv, nnz = inplace_reciprocal(lam)
    VA, lam, inv = VA[:, :nnz], lam[:nnz], inv[:nnz]
    r_ = lam.size
    B = VA * inv[None, :]

    # Phi for this shard (natural order), sample rows overwritten with exact V_A rows
    nloc = hi - lo
    phi = np.empty((nloc, r_), dtype=phi_dtype)
    for s in range(lo, hi, tile):
        idx = np.arange(s, min(s + tile, hi))
        phi[s - lo:s - lo + idx.size] = _affinity_rows(flat, W, idx, sr, sc, sv, sw, pw) @ B
    own = (sel >= lo) & (sel < hi)
    phi[sel[own] - lo] = VA[own]

    def pass_dot(u):
        return np.concatenate([phi[s:s + tile].astype(np.float64) @ u
                               for s in range(0, nloc, tile)]) if nloc else np.zeros(0)

    def pass_tsum(y):
        t = np.zeros(r_)
        for s in range(0, nloc, tile):
            t += phi[s:s + tile].astype(np.float64).T @ y[s:s + tile]
        return allreduce(t)

    t = pass_tsum(np.ones(nloc))
    u_c = u_r = None
    for _ in range(n_sinkhorn_iter):
        u_c = lam * t
        c, _ = inplace_reciprocal(pass_dot(u_c))
        t = pass_tsum(c)
        u_r = lam * t
        rv, _ = inplace_reciprocal(pass_dot(u_r))
        t = pass_tsum(rv)
    if n_sinkhorn_iter == 0:
        raise RuntimeError("streaming oracle needs nSinkhornIter >= 1")

    q = r_
    A = sel[:q]  # pixel indices of the A block
    phiA = VA[:q]
    cA, _ = inplace_reciprocal(phiA @ u_c)
    rA, _ = inplace_reciprocal(phiA @ u_r)
    left = rA[:, None] * (phiA * lam[None, :])
    Wa = left @ (cA[:, None] * phiA).T
    M = left.T  # r x q : diag(lam) PhiA^T diag(rA)

    in_A = np.zeros(nloc, dtype=bool)
    ownA = (A >= lo) & (A < hi)
    in_A[A[ownA] - lo] = True
    c_loc, _ = inplace_reciprocal(pass_dot(u_c))
    if intermediates is not None:
        intermediates["c"] = c_loc.copy()
        intermediates["lam"] = lam.copy()
    c_loc[in_A] = 0.0
    Gm = np.zeros((r_, r_))
    for s in range(0, nloc, tile):
        z = phi[s:s + tile].astype(np.float64) * c_loc[s:s + tile, None]
        Gm += z.T @ z
    Gm = allreduce(Gm)

    cut_info = [] if intermediates is not None else None
    U2, l2 = eigen_decomposition(Wa, info=cut_info)
    ir, _ = inplace_reciprocal(l2)
    S = (U2 * np.sqrt(ir)[None, :]) @ U2.T
    Q = Wa + S @ (M.T @ Gm @ M) @ S
    Vq, Sq = eigen_decomposition(Q, info=cut_info)
    if intermediates is not None:
        intermediates["cuts"] = cut_info
        intermediates["wa_eigvals"] = l2.copy()
        intermediates["q_eigvals"] = Sq.copy()
    k = min(n_eigen_vectors, Vq.shape[1])
    Vq, Sq = Vq[:, :k], Sq[:k]
    irs, _ = inplace_reciprocal(Sq)
    T2 = (S @ Vq) * np.sqrt(irs)[None, :]  # q x K
    Cproj = M @ T2  # r x K
    V = np.empty((nloc, k))
    for s in range(0, nloc, tile):
        V[s:s + tile] = (phi[s:s + tile].astype(np.float64) * c_loc[s:s + tile, None]) @ Cproj
    V[A[ownA] - lo] = (Wa @ T2)[ownA]
    return V, Sq


def slab_rows(H: int, g: int, G: int):
    """Image-row slab owned by shard g of G: rows [g*H//G, (g+1)*H//G)."""
    return (g * H) // G, ((g + 1) * H) // G


def apply_layers_streaming(V_local, eigvals, x_local, n_layers, allreduce=None):
    """Sharded apply: t = allreduce(V_local^T x_local); y_j = V_local (resp_j o t)."""
    if allreduce is None:
        allreduce = lambda x: x
    t = allreduce(V_local.T @ np.asarray(x_local, dtype=np.float64).ravel())
    resp = layer_responses(eigvals, n_layers)
    return np.stack([V_local @ (resp[j] * t) for j in range(n_layers)])


# ------------------------------------------------------------------ synthetic input
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synthetic_luminance(H: int, W: int, seed: int = 1234) -> np.ndarray:
    """SURVEY.md section 8d synthetic input: integer-valued 0..255 (like 8-bit Lab L),
    L = clip(round(128 + 70 s(r/H, c/W) + 40 (u - 1/2))),
    s(a,b) = 1/2 sin(2 pi (1.5a + 0.5b)) + 1/2 cos(2 pi (0.7a - 2.2b)),
    u = splitmix64((r*W+c) xor seed) mapped to [0,1)."""
    r = np.arange(H, dtype=np.float64)[:, None] / H
    c = np.arange(W, dtype=np.float64)[None, :] / W
    s = 0.5 * np.sin(2 * np.pi * (1.5 * r + 0.5 * c)) + 0.5 * np.cos(2 * np.pi * (0.7 * r - 2.2 * c))
    idx = (np.arange(H, dtype=np.uint64)[:, None] * np.uint64(W)
           + np.arange(W, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        h = _splitmix64(idx ^ np.uint64(seed))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return np.clip(np.rint(128.0 + 70.0 * s + 40.0 * (u - 0.5)), 0, 255)


# ------------------------------------------------------------------- colour wrapper
def _srgb_to_linear(c):
    return np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4))


def _linear_to_srgb(c):
    return np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(np.maximum(c, 0), 1 / 2.4) - 0.055)


_XN, _ZN = 0.950456, 1.088754
_M = np.array([[0.412453, 0.357580, 0.180423],
               [0.212671, 0.715160, 0.072169],
               [0.019334, 0.119193, 0.950227]])


# cv::cvtColor(COLOR_BGR2Lab) on 8-bit images is not the float formula of its documentation but a fixed-point table
# implementation (OpenCV imgproc, color_lab.cpp, `RGB2Lab_b`; third-party dependency of the reference, absent here).  Its
# published algorithm, restated: sRGB decode through a 256-entry table scaled by 255 * 2^3, the XYZ matrix in 12-bit fixed
# point with the white point folded in, f(t) through a table of 256 * 3 / 2 * 2^3 entries scaled by 2^15, then
#   L = (296 fY - 1336934 + 2^14) >> 15,   a = (500 (fX - fY) + 128 * 2^15 + 2^14) >> 15,   b = (200 (fY - fZ) + ...) >> 15.
# The tables are made in SINGLE precision there (OpenCV's bit-exact soft-float: t = (1 / 2040)_f32 * i, the cube root by a
# quartic rational polynomial in double whose result is TRUNCATED to 24 bits, the product with 2^15 rounded half to even):
# two of the 3072 entries (49, 628) differ by one from the double-precision table rounds 1-3 used, which changes L
# nowhere on the README images and a or b on ~1.5e-4 of the pixels -- found in round 4, when OpenCV's integer Lab -> BGR
# (below) made the author's FILES reproducible byte for byte and 16 pixels of `flower` were left over.
# Pinned by the reference's own README pairs (tests/test_oracle_readme_pairs.py): with THIS conversion the oracle reproduces
# the author's output images to 0.003 (paper) .. 0.5 grey levels, `bird` to 0.010 -- with the float formula of the
# documentation (`bgr_to_lab8_float`, 14-16 % of the L pixels one level off) bird was 3.35 off, the others 0.2-0.8.
_LAB_SHIFT, _GAMMA_SHIFT = 12, 3
_LAB_SHIFT2 = _LAB_SHIFT + _GAMMA_SHIFT
_F32 = np.float32
# quartic rational approximation of x^(1/3) on [1/8, 1) (OpenCV's `cubeRoot`, error < 2^-24)
_CBRT_P = (45.2548339756803022511987494, 192.2798368355061050458134625, 119.1654824285581628956914143,
           13.43250139086239872172837314, 0.1636161226585754240958355063)
_CBRT_Q = (14.80884093219134573786480845, 151.9714051044435648658557668, 168.5254414101568283957668343,
           33.9905941350215598754191872, 1.0)


def _cube_root_f32(x) -> float:
    """OpenCV's single-precision cube root for a positive float32: mantissa and (exponent mod 3) go through the rational
    polynomial in double, the quotient's mantissa is cut (not rounded) to 23 bits."""
    u = int(np.float32(x).view(np.uint32))
    if u & 0x7FFFFFFF == 0:
        return 0.0
    ex = ((u >> 23) & 0xFF) - 127
    shx = abs(ex) % 3 * (1 if ex >= 0 else -1)        # C's %: sign of the dividend
    if shx >= 0:
        shx -= 3
    ex3 = (ex - shx) // 3                               # exact
    fr = float(np.uint64(((shx + 1023) << 52) | ((u & 0x7FFFFF) << 29)).view(np.float64))   # 1/8 <= fr < 1
    P, Q = _CBRT_P, _CBRT_Q
    fr = ((((P[0] * fr + P[1]) * fr + P[2]) * fr + P[3]) * fr + P[4]) / ((((Q[0] * fr + Q[1]) * fr + Q[2]) * fr + Q[3]) * fr + Q[4])
    fb = int(np.float64(fr).view(np.uint64))
    e = ((fb >> 52) & 0x7FF) - 1023 + ex3 + 127
    return float(np.uint32((e << 23) | ((fb >> 29) & 0x7FFFFF)).view(np.float32))


_tables_cache = {}


def lab8_tables():
    """(gamma[256], cbrt[3072], coeffs[3][3]) of the fixed-point BGR -> Lab conversion"""
    if "fwd" in _tables_cache:
        return _tables_cache["fwd"]
    gamma = np.empty(256, dtype=np.int64)
    for i in range(256):
        xd = float(_F32(i) / _F32(255))
        v = xd / 12.92 if xd <= 0.04045 else ((xd + 0.055) / 1.055) ** 2.4
        gamma[i] = int(np.rint(_F32(255 * (1 << _GAMMA_SHIFT)) * _F32(v)))
    n = 256 * 3 // 2 * (1 << _GAMMA_SHIFT)
    scale = _F32(1) / (_F32(255) * _F32(1 << _GAMMA_SHIFT))
    lthresh, lscale, lbias = _F32(216) / _F32(24389), _F32(841) / _F32(108), _F32(16) / _F32(116)
    cbrt = np.empty(n, dtype=np.int64)
    for i in range(n):
        x = scale * _F32(i)
        v = _F32(float(x) * float(lscale) + float(lbias)) if x < lthresh else _F32(_cube_root_f32(x))
        cbrt[i] = int(np.rint(_F32(1 << _LAB_SHIFT2) * v))
    coeffs = np.rint((1 << _LAB_SHIFT) * _M / np.array([_XN, 1.0, _ZN])[:, None]).astype(np.int64)
    _tables_cache["fwd"] = (gamma, cbrt, coeffs)
    return _tables_cache["fwd"]


def bgr_to_lab8(bgr: np.ndarray) -> np.ndarray:
    """8-bit BGR -> 8-bit Lab as `cv::cvtColor(COLOR_BGR2Lab)` computes it on CV_8UC3 (src/filter.cpp:423,463)."""
    gamma, cbrt, C = lab8_tables()
    descale = lambda v, n: (v + (1 << (n - 1))) >> n
    B, G, R = (gamma[bgr[..., k].astype(np.int64)] for k in range(3))
    fX = cbrt[descale(R * C[0, 0] + G * C[0, 1] + B * C[0, 2], _LAB_SHIFT)]
    fY = cbrt[descale(R * C[1, 0] + G * C[1, 1] + B * C[1, 2], _LAB_SHIFT)]
    fZ = cbrt[descale(R * C[2, 0] + G * C[2, 1] + B * C[2, 2], _LAB_SHIFT)]
    Lscale = (116 * 255 + 50) // 100
    Lshift = -((16 * 255 * (1 << _LAB_SHIFT2) + 50) // 100)
    L = descale(Lscale * fY + Lshift, _LAB_SHIFT2)
    a = descale(500 * (fX - fY) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    b = descale(200 * (fY - fZ) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    return np.clip(np.stack([L, a, b], axis=-1), 0, 255).astype(np.uint8)


def bgr_to_lab8_float(bgr: np.ndarray) -> np.ndarray:
    """The float formula OpenCV DOCUMENTS for 8-bit images (sRGB decode, XYZ D65, L*a*b*, L*255/100, a+128, b+128, rounded):
    what rounds 1-3 used; kept to show the difference (about one grey level on one pixel in seven)."""
    rgb = bgr[..., ::-1].astype(np.float64) / 255.0
    lin = _srgb_to_linear(rgb)
    xyz = lin @ _M.T
    x, y, z = xyz[..., 0] / _XN, xyz[..., 1], xyz[..., 2] / _ZN
    f = lambda t: np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
    L = np.where(y > 0.008856, 116.0 * np.cbrt(y) - 16.0, 903.3 * y)
    a = 500.0 * (f(x) - f(y))
    b = 200.0 * (f(y) - f(z))
    lab = np.stack([L * 255.0 / 100.0, a + 128.0, b + 128.0], axis=-1)
    return np.clip(np.rint(lab), 0, 255).astype(np.uint8)


# cv::cvtColor(COLOR_Lab2BGR) on 8-bit images (src/filter.cpp:440) is likewise an integer algorithm in the OpenCV the author
# used (imgproc `Lab2RGBinteger`, the "bit-exact" path of OpenCV >= 3.4): with BASE = 2^14,
#   (y, fy) = LabToYF[L]                                     two 256-entry tables made in single precision
#   fx = fy + ((5 a 53687 + 2^7) >> 13) - 128 BASE / 500,    fz = fy - (((b 41943 + 2^4) >> 9) - 128 BASE / 200 + 1)
#   x = abToXZ[fx], z = abToXZ[fz]:   t <= 3390 ? t 108 / 841 - BASE 16 / 116 108 / 841 : t t / BASE t / BASE   (C integer division)
#   (r, g, b) = (C (x, y, z) + 2^13) >> 14   with C = round(2^12 XYZ2sRGB_D65 diag(white)),  clamped to 0 .. 4095,
#   each through a 4096-entry sRGB encoding table round(255 gamma^-1(i / 4096)).
# Pinned by the author's output FILES (tests/test_oracle_readme_pairs.py): with this inverse and the single-precision forward
# tables above, `flower` and `brickwall` are reproduced byte for byte (all 3 x 106 800 / 3 x 146 160 values), and on every
# other pair the bytes that differ are explained by the filtered L plane being one level off at a rounding tie of an
# ill-conditioned example (bird: 905 of 182 865 pixels).  With the float formula (`lab8_to_bgr_float`, rounds 1-3) 1.5-7 %
# of the bytes differed.
_INV_BASE_SHIFT = 14
_INV_BASE = 1 << _INV_BASE_SHIFT
_INV_GAMMA_N = 1 << 12
_MIN_AB = -8145
_M_INV = np.array([[3.240479, -1.53715, -0.498535],
                   [-0.969256, 1.875991, 0.041556],
                   [0.055648, -0.204043, 1.057311]])


def _cdiv(a: int, b: int) -> int:
    """C integer division (truncation toward zero)"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def lab8_inverse_tables():
    """(LabToYF[256][2] = (y, fy) scaled by 2^14, abToXZ[36864] indexed from -8145, inv_gamma[4096], coeffs[3][3] rows R, G, B)"""
    if "inv" in _tables_cache:
        return _tables_cache["inv"]
    B = _INV_BASE
    yf = np.empty((256, 2), dtype=np.int64)
    for i in range(256):
        if i <= 20:   # 8 * 255 / 100: the linear piece of L*
            y = int(np.rint(_F32(i * B * 20 * 9) / _F32(17 * 29 * 29 * 29)))
            fy = int(np.rint(_F32(B) * (_F32(16) / _F32(116) + _F32(i * 5) / _F32(3 * 17 * 29))))
        else:
            f = _F32(i * 100 * B) / _F32(255 * 116) + _F32(16 * B) / _F32(116)
            fy = int(np.rint(f))
            y = int(np.rint(f * f * f / _F32(B * B)))
        yf[i] = (y, fy)
    n = B * 9 // 4
    ab = np.empty(n, dtype=np.int64)
    off = _cdiv(_cdiv(B * 16, 116) * 108, 841)
    for t in range(_MIN_AB, n + _MIN_AB):
        ab[t - _MIN_AB] = _cdiv(t * 108, 841) - off if t <= 3390 else _cdiv(_cdiv(t * t, B) * t, B)
    ig = np.empty(_INV_GAMMA_N, dtype=np.int64)
    for i in range(_INV_GAMMA_N):
        xd = float(_F32(i) * (_F32(1) / _F32(_INV_GAMMA_N)))
        v = xd * 12.92 if xd <= 0.0031308 else xd ** (1.0 / 2.4) * 1.055 - 0.055
        ig[i] = int(np.rint(_F32(255) * _F32(v)))
    coeffs = np.rint((1 << _LAB_SHIFT) * _M_INV * np.array([_XN, 1.0, _ZN])[None, :]).astype(np.int64)
    _tables_cache["inv"] = (yf, ab, ig, coeffs)
    return _tables_cache["inv"]


def lab8_to_bgr(lab: np.ndarray) -> np.ndarray:
    """8-bit Lab -> 8-bit BGR as `cv::cvtColor(COLOR_Lab2BGR)` computes it on CV_8UC3 (src/filter.cpp:440)."""
    yf, ab, ig, C = lab8_inverse_tables()
    L, a, b = (lab[..., k].astype(np.int64) for k in range(3))
    y, fy = yf[L, 0], yf[L, 1]
    adiv = ((5 * a * 53687 + (1 << 7)) >> 13) - 128 * _INV_BASE // 500
    bdiv = ((b * 41943 + (1 << 4)) >> 9) - 128 * _INV_BASE // 200 + 1
    x, z = ab[fy + adiv - _MIN_AB], ab[fy - bdiv - _MIN_AB]
    sh = _LAB_SHIFT + _INV_BASE_SHIFT - 12
    out = [ig[np.clip((C[k, 0] * x + C[k, 1] * y + C[k, 2] * z + (1 << (sh - 1))) >> sh, 0, _INV_GAMMA_N - 1)] for k in (2, 1, 0)]
    return np.stack(out, axis=-1).astype(np.uint8)


def lab8_to_bgr_float(lab: np.ndarray) -> np.ndarray:
    """The float formula OpenCV DOCUMENTS for Lab -> BGR: what rounds 1-3 used; kept to show the difference (1.5-7 % of the
    bytes of an output file)."""
    L = lab[..., 0].astype(np.float64) * 100.0 / 255.0
    a = lab[..., 1].astype(np.float64) - 128.0
    b = lab[..., 2].astype(np.float64) - 128.0
    fy = (L + 16.0) / 116.0
    y = np.where(L > 7.9996248, fy ** 3, L / 903.3)
    fy = np.where(L > 7.9996248, fy, 7.787 * y + 16.0 / 116.0)
    fx = a / 500.0 + fy
    fz = fy - b / 200.0
    finv = lambda t: np.where(t > 0.206893, t ** 3, (t - 16.0 / 116.0) / 7.787)
    xyz = np.stack([finv(fx) * _XN, y, finv(fz) * _ZN], axis=-1)
    lin = xyz @ np.linalg.inv(_M).T
    rgb = _linear_to_srgb(np.clip(lin, 0, 1))
    return np.clip(np.rint(rgb[..., ::-1] * 255.0), 0, 255).astype(np.uint8)


def enhance_image(bgr: np.ndarray, n_row_samples, n_col_samples, hx, hy,
                  n_sinkhorn_iter, n_eigen_vectors, weights, info: list | None = None,
                  force_rank: int | None = None, return_L: bool = False):
    """`trainForEnhancement` + `enhance`, src/filter.cpp:514-519, 412-443.  With `return_L` also returns the
    8-bit filtered L plane (before Lab -> BGR) and the eigenvalues."""
    if bgr.ndim != 3 or bgr.shape[2] != 3:
        raise RuntimeError("Can only enhance RGB image.")  # :414-416
    lab = bgr_to_lab8(bgr)
    L = lab[..., 0].astype(np.float64)
    V, S = train_filter(L, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter, n_eigen_vectors,
                        info=info, force_rank=force_rank)
    y = apply_filter(V, L, transform_eigenvalues(S, weights))
    y = np.clip(y, 0, 255)  # :434-435
    lab2 = lab.copy()
    lab2[..., 0] = np.rint(y).astype(np.uint8)  # convertTo(CV_8U): round-half-even, :436
    out = lab8_to_bgr(lab2)
    if return_L:
        return out, lab2[..., 0].copy(), S
    return out


# --------------------------------------------------------------------------- denoise wrapper
def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    """cv::BORDER_DEFAULT = BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba"""
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * n - 2
    i = np.mod(idx, period)
    return np.where(i < n, i, period - i)


def bilateral_tables(sigma_color: float, sigma_space: float):
    """radius and the two fp32 weight tables of cv::bilateralFilter (CV_8UC1, d <= 0), as OpenCV documents them.
    OpenCV itself is not in this image: third-party arithmetic restated from its documentation, unpinned."""
    if sigma_color <= 0:
        sigma_color = 1.0
    if sigma_space <= 0:
        sigma_space = 1.0
    cc, sc = -0.5 / (sigma_color * sigma_color), -0.5 / (sigma_space * sigma_space)
    radius = max(int(np.rint(sigma_space * 1.5)), 1)
    colour_w = np.exp((np.arange(256, dtype=np.float64) ** 2) * cc).astype(np.float32)
    ii, jj = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    rr = np.sqrt((ii * ii + jj * jj).astype(np.float64))
    space_w = np.where(rr > radius, 0.0, np.exp(rr * rr * sc)).astype(np.float32)
    return radius, space_w, colour_w


def bilateral8(plane: np.ndarray, sigma_color: float, sigma_space: float) -> np.ndarray:
    """cv::bilateralFilter(src, dst, -1, sigmaColor, sigmaSpace, BORDER_DEFAULT) on one 8-bit channel, the call of
    src/filter.cpp:371 and :535.  fp32 products and sums in row-major window order (no fused multiply-add),
    round-half-even of sum / wsum.  `plane`: integers 0..255 (any dtype); returns uint8."""
    src = np.asarray(plane).astype(np.int64)
    H, W = src.shape
    radius, space_w, colour_w = bilateral_tables(sigma_color, sigma_space)
    rows = _reflect101(np.arange(-radius, H + radius), H)
    cols = _reflect101(np.arange(-radius, W + radius), W)
    pad = src[np.ix_(rows, cols)]
    v0 = src
    s = np.zeros((H, W), dtype=np.float32)
    ws = np.zeros((H, W), dtype=np.float32)
    d = 2 * radius + 1
    for dy in range(d):
        for dx in range(d):
            sw = space_w[dy, dx]
            if sw == 0.0:
                continue
            v = pad[dy:dy + H, dx:dx + W]
            w = (sw * colour_w[np.abs(v - v0)]).astype(np.float32)          # fp32 product
            s = (s + (v.astype(np.float32) * w).astype(np.float32)).astype(np.float32)
            ws = (ws + w).astype(np.float32)
    return np.rint((s / ws).astype(np.float32)).astype(np.uint8)


def denoise_image(bgr: np.ndarray, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter, n_eigen_vectors,
                  sigma_color: int, sigma_space: int, k: float) -> np.ndarray:
    """`trainForDenoise` + `denoise`, src/filter.cpp:521-538, 349-410: the filter is trained on the bilateral-filtered
    L channel; a and b go through `apply` with the eigenvalues shrunk to min(lambda, 1)^k; L becomes the
    bilateral-filtered plane (the `apply` on channel 0 is commented out, :389)."""
    if bgr.ndim != 3 or bgr.shape[2] != 3:
        raise RuntimeError("Can only enchance RGB image.")  # :351-353 (sic)
    lab = bgr_to_lab8(bgr)
    Y = bilateral8(lab[..., 0], sigma_color, sigma_space)
    V, S = train_filter(Y.astype(np.float64), n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter, n_eigen_vectors)
    t = np.minimum(S, 1.0) ** k  # :381-387
    out = lab.copy()
    out[..., 0] = Y
    for ch in (1, 2):
        y = apply_filter(V, lab[..., ch].astype(np.float64), t).reshape(lab.shape[:2])
        out[..., ch] = np.rint(np.clip(y, 0, 255)).astype(np.uint8)  # :394-399
    return lab8_to_bgr(out)
