// The orthogonalisation stage (reference src/filter.cpp:282-331: W blocks, W_A's inverse root, Q, its leading eigenpairs,
// the projection coefficients) in its three forms: on the host from materialised quantities, in sample space on the host,
// in sample space with the p x p products and solvers on the device.  Split out of pipeline.hip (round 3).
#include "ortho.h"

using namespace nlep;

namespace {

// ---- small host algebra, column-major ----
// The p x p products of the orthogonalisation run on the register-blocked kernels of eigen_sym.cpp (a 200^3
// product is ~0.2 ms on one core); only products of more than ~80 MFLOP per thread are split by output columns
// over short-lived threads (nleh::run_parts; thread start-up and remote caches cost more than that on the GPU box).
template <typename F>
void par_cols(int n, long long work_per_col, F&& body) {
    const long long work = work_per_col * n;  // multiply-adds
    int nt = (int)std::min<long long>(16, work / 40000000 + 1);
    if (const char* e = std::getenv("NLE_HOST_THREADS")) nt = std::max(1, std::atoi(e));
    nt = std::min(nt, n);
    if (nt <= 1) {
        body(0, n);
        return;
    }
    nleh::run_parts(nt, nt, [&](int t) { body((int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt)); });
}
// C (m x n) = A (m x k) * B (k x n)
void mm(const double* A, const double* B, double* C, int m, int k, int n) {
    par_cols(n, (long long)m * k, [&](int j0, int j1) { nleh::gemm_nn_cols(A, B, C, m, k, n, j0, j1); });
}
// C (m x n) = A (m x k) * B^T (B is n x k)
void mm_nt(const double* A, const double* B, double* C, int m, int k, int n) {
    par_cols(n, (long long)m * k, [&](int j0, int j1) { nleh::gemm_nt_cols(A, B, C, m, k, n, j0, j1); });
}
// C (k x n) = A^T (A is m x k) * B (m x n)
void mm_tn(const double* A, const double* B, double* C, int m, int k, int n) {
    par_cols(n, (long long)m * k, [&](int j0, int j1) { nleh::gemm_tn_cols(A, B, C, m, k, n, j0, j1); });
}

// Top eigenpairs of Q (reference src/filter.cpp:310-317).  solver 0: the default build's eigenDecomposition(Q) -- all
// eigenvalues, the leading run >= eps counted in *rq, eigenvectors of the first min(n_eig, q) only.  solver 1: the
// USE_SPECTRA build's topkEigenDecomposition (:170-199): nev = min(n_eig, q - 1) pairs of largest magnitude by Lanczos,
// *rq = converged pairs in the leading run >= eps.  Vq: q x (columns formed), Sq: their eigenvalues, descending.
void top_eigenpairs(const std::vector<double>& Qm, int q, int n_eig, int solver, std::vector<double>* Vq,
                    std::vector<double>* Sq, int* rq) {
    if (solver == 1 && q > 1) {
        const int nev = std::min(std::max(n_eig, 1), q - 1);
        Vq->assign((size_t)q * nev, 0.0);
        Sq->assign(nev, 0.0);
        int restarts = 0;
        const int nconv = nleh::lanczos_topk(Qm.data(), q, nev, NLE_EPS, 1000, Vq->data(), Sq->data(), &restarts);
        if (nconv < 0) throw Fail{NLE_ERR_NUMERIC, "Lanczos: the projected eigenproblem did not converge"};
        if (nconv < nev)  // Spectra only warns on stderr and goes on with the converged pairs (:180-183)
            std::fprintf(stderr, "# converged eigenvalues: %d\nEigen decomposition NOT successful. Results might be inaccurate.\n", nconv);
        int r = 0;
        while (r < nconv && (*Sq)[r] >= NLE_EPS) ++r;  // :186-196
        *rq = r;
        if (std::getenv("NLE_TRACE")) std::fprintf(stderr, "[nle trace] Lanczos top-%d of %d: %d restarts, %d converged\n", nev, q, restarts, nconv);
        return;
    }
    Vq->assign((size_t)q * std::min(q, std::max(n_eig, 1)), 0.0);  // only the kept eigenvectors (:314)
    Sq->assign(q, 0.0);  // the leading min(n_eig, q) eigenvalues; the count of the cut comes back in *rq (:313-316 use no more)
    if (!nleh::eigen_decomposition_topk(Qm.data(), q, NLE_EPS, n_eig, Vq->data(), Sq->data(), rq))
        throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge on Q"};
}

}  // namespace

// The W blocks and the orthogonalisation on the host.  Inputs: V_A (p x r), lambda, u_c, u_r,
// G = sum over ALL pixels of c^2 phi phi^T.  Outputs: Sq (K'), Cproj (r x K'), VArows (q x K').
// reference src/filter.cpp:247-250 (W blocks, q = phi.cols()), :282-331 (orthogonalize).
namespace nlep {

Ortho orthogonalize_host(const Nystrom& ny, int p, const std::vector<double>& u_c,
                         const std::vector<double>& u_r, std::vector<double> G, int n_eig, bool device_f32,
                         int topk_solver) {
    const int r = ny.r, q = ny.r;  // :247 -- the A block is the first q = r permuted rows
    // phi_A = V_A[:q] (exact, fp64); what the device holds for those rows is float(V_A)
    std::vector<double> cA(q), rA(q), cA32(q);
    std::vector<double> left((size_t)q * r), right((size_t)q * r), phi32((size_t)q * r);
    for (int a = 0; a < q; ++a) {
        double sc = 0.0, sr = 0.0, sc32 = 0.0;
        for (int k = 0; k < r; ++k) {
            const double v = ny.VA[(size_t)k * p + a];
            const double v32 = device_f32 ? (double)(float)v : v;  // what the device holds for that row
            phi32[(size_t)k * q + a] = v32;
            sc += v * u_c[k];
            sr += v * u_r[k];
            sc32 += v32 * u_c[k];
        }
        cA[a] = recip0(sc);
        rA[a] = recip0(sr);
        cA32[a] = recip0(sc32);
    }
    for (int k = 0; k < r; ++k)
        for (int a = 0; a < q; ++a) {
            const double v = ny.VA[(size_t)k * p + a];
            left[(size_t)k * q + a] = rA[a] * v * ny.lam[k];  // R * (phi_top * D)
            right[(size_t)k * q + a] = cA[a] * v;             // c o phi_top
        }
    Ortho o;
    o.q = q;
    o.Wa.resize((size_t)q * q);
    mm_nt(left.data(), right.data(), o.Wa.data(), q, r, q);  // :249
    // remove the A rows from the all-pixel Gram: G_B = G - sum_a c_a^2 phi_a phi_a^T
    for (int a = 0; a < q; ++a) {
        const double c2 = cA32[a] * cA32[a];
        for (int j = 0; j < r; ++j) {
            const double vj = c2 * phi32[(size_t)j * q + a];
            for (int i = 0; i < r; ++i) G[(size_t)j * r + i] -= phi32[(size_t)i * q + a] * vj;
        }
    }
    // Wab Wab^T = left * G_B * left^T  (:296)
    std::vector<double> LG((size_t)q * r), WW((size_t)q * q);
    mm(left.data(), G.data(), LG.data(), q, r, r);
    mm_nt(LG.data(), left.data(), WW.data(), q, r, q);
    // S = Wa^{-1/2} (pseudo-inverse root), :287-292
    std::vector<double> U2((size_t)q * q), l2(q);
    int r2 = 0;
    if (!nleh::eigen_decomposition(o.Wa.data(), q, NLE_EPS, U2.data(), l2.data(), &r2))
        throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge on Wa"};
    std::vector<double> Us((size_t)q * std::max(r2, 1)), S((size_t)q * q);
    for (int k = 0; k < r2; ++k) {
        const double s = std::sqrt(recip0(l2[k]));
        for (int i = 0; i < q; ++i) Us[(size_t)k * q + i] = U2[(size_t)k * q + i] * s;
    }
    o.r_wa = r2;
    if (r2 <= 0) throw Fail{NLE_ERR_NUMERIC, "Wa has no eigenvalue >= 1e-10"};
    std::vector<double> Vq, Sq, T2;
    int rq = 0, K = 0;
    if (topk_solver == 0) {
        // Q = Wa + S (Wab Wab^T) S  (:296) on the subspace the Wa cut kept: with F = U2 L2^-1/2 (q x r2), S = F U2^T and
        // Q = U2 (L2 + F^T WW F) U2^T + U1 L1 U1^T, so the eigenpairs of Q are those of Qt = L2 + F^T WW F (order r2)
        // mapped by U2, plus the dropped (U1, L1 < 1e-10) -- which the literal S (.) S product, with entries of S up to
        // 1e5, can lift back over the cut by its rounding alone (tools/parity_fuzz.py, seed 12 "big", case 59: 9.5e-11
        // became 1.11e-10 here and 9.97e-11 in numpy).  S Vq = F Vt.
        std::vector<double> WF((size_t)q * r2), Qt((size_t)r2 * r2);
        mm(WW.data(), Us.data(), WF.data(), q, q, r2);
        mm_tn(Us.data(), WF.data(), Qt.data(), q, r2, r2);
        for (int k = 0; k < r2; ++k) Qt[(size_t)k * r2 + k] += l2[k];
        std::vector<double> Vt;
        top_eigenpairs(Qt, r2, n_eig, topk_solver, &Vt, &Sq, &rq);
        K = std::min(n_eig, rq);  // :314
        if (K <= 0) throw Fail{NLE_ERR_NUMERIC, "Q has no eigenvalue >= 1e-10"};
        T2.resize((size_t)q * K);
        mm(Us.data(), Vt.data(), T2.data(), q, r2, K);
    } else {
        // the USE_SPECTRA build's solver works on the literal Q (full matrix, :170-199)
        mm_nt(Us.data(), U2.data(), S.data(), q, r2, q);
        std::vector<double> T1((size_t)q * q), Qm((size_t)q * q);
        mm(S.data(), WW.data(), T1.data(), q, q, q);
        mm(T1.data(), S.data(), Qm.data(), q, q, q);
        for (size_t i = 0; i < Qm.size(); ++i) Qm[i] += o.Wa[i];
        top_eigenpairs(Qm, q, n_eig, topk_solver, &Vq, &Sq, &rq);
        K = std::min(n_eig, rq);  // :314
        if (K <= 0) throw Fail{NLE_ERR_NUMERIC, "Q has no eigenvalue >= 1e-10"};
        T2.resize((size_t)q * K);
        mm(S.data(), Vq.data(), T2.data(), q, q, K);
    }
    o.K = K;
    o.r_q = rq;
    o.Sq.assign(Sq.begin(), Sq.begin() + K);
    // T2 = S * Vq * diag(Sq^-1/2)  (q x K)
    for (int k = 0; k < K; ++k) {
        const double s = std::sqrt(recip0(Sq[k]));
        for (int i = 0; i < q; ++i) T2[(size_t)k * q + i] *= s;
    }
    o.Cproj.resize((size_t)r * K);
    mm_tn(left.data(), T2.data(), o.Cproj.data(), q, r, K);  // M * T2, M = left^T
    o.VArows.resize((size_t)q * K);
    mm(o.Wa.data(), T2.data(), o.VArows.data(), q, q, K);  // top block of :327
    return o;
}

// first half: everything that does not depend on the Gram matrix (runs on the host while the GPU
// computes Gk): sample scalings, Kr, P, Wa and S = Wa^-1/2 (:287-292)
void ortho_ss_prepare(OrthoSS& o, const Nystrom& ny, int p, const std::vector<double>& sA_c,
                      const std::vector<double>& sA_r, bool literal_q) {
    const int r = ny.r, q = ny.r;
    o.p = p;
    o.r = r;
    o.q = q;
    o.cA.resize(p);
    o.rA.resize(p);
    for (int a = 0; a < p; ++a) {  // sA = V_A u: the samples' row sums under the two final scalings
        o.cA[a] = recip0(sA_c[a]);
        o.rA[a] = recip0(sA_r[a]);
    }
    if (ny.chol) {
        o.Kr = ny.Ka;  // r == p: Kr = Ka, P = I
    } else {
        std::vector<double> VL((size_t)p * r);
        o.Kr.resize((size_t)p * p);
        for (int k = 0; k < r; ++k)
            for (int a = 0; a < p; ++a) VL[(size_t)k * p + a] = ny.VA[(size_t)k * p + a] * ny.lam[k];
        mm_nt(VL.data(), ny.VA.data(), o.Kr.data(), p, r, p);
        if (r < p) {
            o.P.resize((size_t)p * p);
            mm_nt(ny.VA.data(), ny.VA.data(), o.P.data(), p, r, p);
        }
    }
    o.Wa.resize((size_t)q * q);
    for (int b = 0; b < q; ++b)
        for (int a = 0; a < q; ++a) o.Wa[(size_t)b * q + a] = o.rA[a] * o.Kr[(size_t)b * p + a] * o.cA[b];  // :249
    // S with S S^T = A^-1, A = the symmetric matrix the reference's solver sees (lower triangle of Wa).
    // The reference takes the symmetric root A^-1/2 (:287-292); any other root F gives the similar matrix
    // G^T Q G (G = A^1/2 F orthogonal) with the same eigenvalues and the same product S Vq, hence the same
    // eigenvectors V (:327).  When A is provably free of eigenvalues below the cut, F = L^-T (Cholesky).
    o.S.resize((size_t)q * q);
    // (A^-1)_ii >= 1 / A_ii, so sum_i 1 / A_ii above the certificate's bound already rules the Cholesky form out
    double inv_diag = 0.0;
    for (int a = 0; a < q; ++a) inv_diag += o.Wa[(size_t)a * q + a] > 0.0 ? 1.0 / o.Wa[(size_t)a * q + a] : 1e300;
    // (literal_q: the caller wants Q itself, S = Wa^-1/2 the symmetric root -- the Lanczos option -- not a similar matrix)
    if (!literal_q && std::getenv("NLE_FORCE_EIG") == nullptr && inv_diag <= kCholMaxInvTrace) {
        std::vector<double> L((size_t)q * q), Li((size_t)q * q);
        double inv_trace = 0.0;
        if (nleh::cholesky_with_inverse(o.Wa.data(), q, L.data(), Li.data(), &inv_trace, kCholMaxInvTrace) && inv_trace <= kCholMaxInvTrace) {
            std::vector<double> Lt((size_t)q * q);
            for (int k = 0; k < q; ++k)
                for (int a = 0; a < q; ++a) {
                    o.S[(size_t)k * q + a] = Li[(size_t)a * q + k];  // L^-T
                    Lt[(size_t)k * q + a] = L[(size_t)a * q + k];
                }
            o.St = std::move(Li);
            o.A2.resize((size_t)q * q);
            mm(Lt.data(), L.data(), o.A2.data(), q, q, q);  // F^T A^2 F = L^T L
            o.r_wa = q;
            o.chol_wa = true;
            return;
        }
    }
    std::vector<double> U2((size_t)q * q), l2(q);
    int r2 = 0;
    if (!nleh::eigen_decomposition(o.Wa.data(), q, NLE_EPS, U2.data(), l2.data(), &r2))
        throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge on Wa"};
    std::vector<double> Us((size_t)q * std::max(r2, 1));
    for (int k = 0; k < r2; ++k) {
        const double sv = std::sqrt(recip0(l2[k]));
        for (int i = 0; i < q; ++i) Us[(size_t)k * q + i] = U2[(size_t)k * q + i] * sv;
    }
    mm_nt(Us.data(), U2.data(), o.S.data(), q, r2, q);  // :287-292
    o.r_wa = r2;
    if (std::getenv("NLE_TRACE"))
        fprintf(stderr, "[nle trace] Wa: %d of %d eigenvalues >= 1e-10 (largest %.3e, smallest kept %.3e)\n", r2, q, l2[0],
                r2 > 0 ? l2[r2 - 1] : 0.0);
    o.St = o.S;
    o.A2 = o.Wa;  // :296 (the solver reads the lower triangle of the sum)
}

// second half: needs Gk
void ortho_ss_finish(OrthoSS& o, std::vector<double> Gk, int n_eig, int topk_solver) {
    const int p = o.p, r = o.r, q = o.q;
    const std::vector<double>&cA = o.cA, &rA = o.rA, &Kr = o.Kr, &Wa = o.Wa, &S = o.S;
    for (int a = q; a < p; ++a) {  // B-block samples
        const double c2 = cA[a] * cA[a];
        const double* ka = Kr.data() + (size_t)a * p;
        for (int j = 0; j < p; ++j) {
            const double vj = c2 * ka[j];
            for (int i = 0; i < p; ++i) Gk[(size_t)j * p + i] += ka[i] * vj;
        }
    }
    if (r < p) {
        std::vector<double> T((size_t)p * p);
        mm(o.P.data(), Gk.data(), T.data(), p, p, p);
        mm(T.data(), o.P.data(), Gk.data(), p, p, p);
    }
    std::vector<double> WW((size_t)q * q);
    for (int b = 0; b < q; ++b)
        for (int a = 0; a < q; ++a) WW[(size_t)b * q + a] = rA[a] * Gk[(size_t)b * p + a] * rA[b];  // Wab Wab^T, :296
    std::vector<double> T1((size_t)q * q), Qm((size_t)q * q);
    mm(o.St.data(), WW.data(), T1.data(), q, q, q);
    mm(T1.data(), S.data(), Qm.data(), q, q, q);
    for (size_t i = 0; i < Qm.size(); ++i) Qm[i] += o.A2[i];  // :296
    std::vector<double> Vq, Sq;
    int rq = 0;
    top_eigenpairs(Qm, q, n_eig, topk_solver, &Vq, &Sq, &rq);
    const int K = std::min(n_eig, rq);  // :314
    if (K <= 0) throw Fail{NLE_ERR_NUMERIC, "Q has no eigenvalue >= 1e-10"};
    o.K = K;
    o.r_q = rq;
    o.Sq.assign(Sq.begin(), Sq.begin() + K);
    std::vector<double> T2((size_t)q * K), RT2((size_t)q * K);
    mm(S.data(), Vq.data(), T2.data(), q, q, K);
    for (int k = 0; k < K; ++k) {
        const double sv = std::sqrt(recip0(Sq[k]));  // :319-321
        for (int i = 0; i < q; ++i) {
            T2[(size_t)k * q + i] *= sv;
            RT2[(size_t)k * q + i] = rA[i] * T2[(size_t)k * q + i];
        }
    }
    o.D.resize((size_t)p * K);
    if (r < p) {
        mm(o.P.data(), RT2.data(), o.D.data(), p, q, K);  // first q columns of P
    } else {
        o.D = RT2;
    }
    o.Vrows.assign((size_t)p * K, 0.0);
    std::vector<double> WT((size_t)q * K);
    mm(Wa.data(), T2.data(), WT.data(), q, q, K);  // top block of :327
    for (int k = 0; k < K; ++k) {
        for (int a = 0; a < q; ++a) o.Vrows[(size_t)k * p + a] = WT[(size_t)k * q + a];
        for (int a = q; a < p; ++a) {
            double sv = 0.0;
            for (int j = 0; j < p; ++j) sv += Kr[(size_t)a * p + j] * o.D[(size_t)k * p + j];
            o.Vrows[(size_t)k * p + a] = cA[a] * sv;
        }
    }
}

// The same orthogonalisation with every p- and q-sized product on the GPU (generic64.hip: gemm64s, fp64 MFMA); the host
// keeps what is inherently serial -- the two symmetric eigensolves (Wa, Q) or their Cholesky shortcut.  `d_Gk`: the local
// Gram matrix (p x p, device); `enqueue_gram` puts the Gram kernels on the stream (they run under the host's eigensolve of
// Wa), `reduce_gram` sums d_Gk over the ranks.  On return o.K, o.Sq, o.D, o.Vrows (host, column-major p x K), o.r_wa,
// o.r_q, o.chol_wa are set exactly as ortho_ss_prepare + ortho_ss_finish set them.
// At cfg4 (p = 200) this takes ~0.6 ms of 200^3 host products off the critical path, at cfg5 (p = 900) ~50 ms.
void ortho_ss_device(nle_ctx* c, OrthoSS& o, const Nystrom& ny, int p, const std::vector<double>& sA_c,
                     const std::vector<double>& sA_r, double* d_Gk, int n_eig, const std::function<void()>& enqueue_gram,
                     const std::function<void()>& reduce_gram, double* host_ms, double* host_overlapped_ms, Trace& tr) {
    const int r = ny.r, q = ny.r;
    hipStream_t st = c->stream;
    o.p = p;
    o.r = r;
    o.q = q;
    o.cA.resize(p);
    o.rA.resize(p);
    for (int a = 0; a < p; ++a) {
        o.cA[a] = recip0(sA_c[a]);
        o.rA[a] = recip0(sA_r[a]);
    }
    const size_t pp = (size_t)p * p, qq = (size_t)q * q;
    DevBuf<double> d_rA(p), d_cA(p), d_Kr, d_P, d_VA, d_lam;
    HIP_OK(hipMemcpyAsync(d_rA.p, o.rA.data(), p * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_cA.p, o.cA.data(), p * sizeof(double), hipMemcpyHostToDevice, st));
    if (ny.chol) {
        o.Kr = ny.Ka;  // r == p: Kr = Ka, P = I
    } else {  // Kr = V_r L V_r^T, P = V_r V_r^T on the device; Kr comes back for Wa
        d_VA.alloc((size_t)p * r);
        d_lam.alloc(r);
        d_Kr.alloc(pp);
        HIP_OK(hipMemcpyAsync(d_VA.p, ny.VA.data(), (size_t)p * r * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_lam.p, ny.lam.data(), r * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_OK(nlek::gemm64s(st, p, p, r, d_VA.p, 1, p, d_VA.p, p, 1, d_Kr.p, 1, p, nullptr, d_lam.p));
        if (r < p) {
            d_P.alloc(pp);
            HIP_OK(nlek::gemm64s(st, p, p, r, d_VA.p, 1, p, d_VA.p, p, 1, d_P.p, 1, p));
        }
        o.Kr.resize(pp);
        HIP_OK(hipMemcpyAsync(o.Kr.data(), d_Kr.p, pp * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    // ---- Wa and a factor F of its (pseudo-)inverse, beside the Gram kernels: on the host below dev_solver_min_n(), else on
    // the device on the ctx's second stream (the Gram kernels are on the first)
    double h0 = now_ms();
    const bool dev_wa = use_dev_solver(q) && !std::getenv("NLE_HOST_WA");
    // Wa = diag(rA) Kr[:q,:q] diag(cA) (:249).  On the device route with Ka resident there (solve_Ka's device Cholesky) it is
    // formed on the device; the host copy is built only if the host root has to take over.
    const bool wa_on_device = dev_wa && ny.dev != nullptr;
    auto build_Wa_host = [&] {
        if (!o.Wa.empty()) return;
        o.Wa.resize(qq);
        for (int b = 0; b < q; ++b)
            for (int a = 0; a < q; ++a) o.Wa[(size_t)b * q + a] = o.rA[a] * o.Kr[(size_t)b * p + a] * o.cA[b];  // :249
    };
    if (!wa_on_device) build_Wa_host();
    std::vector<double> L, Li, U2, Us, l2_kept;
    int r2 = 0;
    bool chol_wa = false;
    double inv_diag = 0.0;
    if (!wa_on_device)
        for (int a = 0; a < q; ++a) inv_diag += o.Wa[(size_t)a * q + a] > 0.0 ? 1.0 / o.Wa[(size_t)a * q + a] : 1e300;
    bool deflated = false;
    std::vector<double> Fdefl, Gdefl;  // F (q x q) and G = L^T Vd (q x d):  F^T A^2 F = L^T L - G G^T
    int nd = 0;
    const int max_defl = q / 8;
    const bool force_eig = std::getenv("NLE_FORCE_EIG") != nullptr;
    // device form: the same three routes (Cholesky when the cut provably removes nothing; deflated Cholesky when it removes
    // at most q / 8 eigenvalues; else all kept eigenvectors -- that last one stays on the host)
    DevBuf<double> d_Wa(qq), d_F, d_L, d_G;
    DevSymEig esw;
    DevChol chw;
    DevBuf<double> d_Ah, d_Vd, d_wgt, d_W, d_neg1;
    bool dev_done = false;
    if (dev_wa) {
        hipStream_t sa = aux_stream(c);
        // (workspace taken from the ctx cache now, while the first stream is idle: blocks of the cache are only ordered on it)
        esw.prepare(c, q, sa);
        chw.prepare(c, q, sa);
        d_Ah.alloc(qq);
        d_F.alloc(qq);
        d_Vd.alloc((size_t)q * (max_defl + 1));
        d_wgt.alloc(max_defl + 1);
        d_W.alloc((size_t)(max_defl + 1) * q);
        d_G.alloc((size_t)q * (max_defl + 1));
        d_neg1.alloc(max_defl + 1);
        HIP_OK(hipStreamSynchronize(st));
        const bool wa_serial = std::getenv("NLE_WA_SERIAL") != nullptr;  // measurement: the Gram kernels after the root
        if (!wa_serial) enqueue_gram();
        if (wa_on_device)  // Wa(a, b) = rA[a] Ka(a, b) cA[b]: one product with both diagonals, no upload
            HIP_OK(nlek::scale_rc64(sa, q, ny.dev->Ka.p, p, d_rA.p, d_cA.p, d_Wa.p));
        else
            upload_staged(c, d_Wa.p, o.Wa.data(), qq, sa);
        HIP_OK(nlek::symm_lower64(sa, q, d_Wa.p, d_Ah.p));
        // (no separate Cholesky attempt here: the host's stops at the first pivot that proves it futile, a device
        // factorisation costs as much as the reduction -- so the eigenvalues come first and decide; none below the cut
        // is the deflated route with nothing to deflate)
        bool reduced = false;
        if (!force_eig && std::getenv("NLE_NO_DEFLATE") == nullptr) {
            reduced = esw.reduce(c, q, d_Ah.p, nullptr);
            // a rank whose device reduction failed takes the host root below; its peers must not wait for it in a different
            // sequence of collectives nor cut at another rank: if one falls back, all do (one 8-byte all-reduce, world > 1)
            if (c->world > 1) reduced = ranks_where(c, !reduced) == 0;
            tr.mark("ss:   Wa: tridiagonal form + eigenvalues (device)");
        }
        if (reduced) {
            int kept = 0;
            while (kept < q && esw.D[kept] >= NLE_EPS) ++kept;  // :213-216
            nd = q - kept;
            if (kept > 0 && nd <= max_defl && esw.D[0] > 0.0) {
                const double sig = esw.D[0];
                if (nd > 0) {
                    esw.vectors(c, kept, nd, d_Vd.p);
                    tr.mark("ss:   Wa: dropped eigenvectors (host inverse iteration)");
                    std::vector<double> wgt(nd);
                    for (int t = 0; t < nd; ++t) wgt[t] = sig - esw.D[kept + t];
                    HIP_OK(hipMemcpyAsync(d_wgt.p, wgt.data(), nd * sizeof(double), hipMemcpyHostToDevice, sa));
                    // Ahat = A + Vd (sig I - Ld) Vd^T
                    HIP_OK(nlek::gemm64s(sa, q, q, nd, d_Vd.p, 1, q, d_Vd.p, q, 1, d_Ah.p, 1, q, nullptr, d_wgt.p, nullptr, d_Ah.p, 1, q));
                    HIP_OK(hipStreamSynchronize(sa));  // `wgt` (host) is consumed
                }
                chw.factor(c, q, d_Ah.p);
                const bool fact_ok = chw.finish(c);
                if (!fact_ok && std::getenv("NLE_TRACE"))
                    fprintf(stderr, "[nle trace] Wa: the deflated matrix did not factor (trace of the inverse %.3e, %d dropped, sigma %.3e)\n",
                            chw.inv_trace, nd, sig);
                if (fact_ok) {
                    // F = L^-T - Vd (Vd^T L^-T),  G = L^T Vd
                    HIP_OK(nlek::transpose64(sa, q, chw.Linv.p, d_F.p));
                    if (nd > 0) {
                        HIP_OK(nlek::fill64(sa, d_neg1.p, nd, -1.0));
                        HIP_OK(nlek::gemm64s(sa, nd, q, q, d_Vd.p, q, 1, chw.Linv.p, q, 1, d_W.p, 1, nd));
                        HIP_OK(nlek::gemm64s(sa, q, q, nd, d_Vd.p, 1, q, d_W.p, 1, nd, d_F.p, 1, q, nullptr, d_neg1.p, nullptr, d_F.p, 1, q));
                        HIP_OK(nlek::gemm64s(sa, q, nd, q, chw.L.p, q, 1, d_Vd.p, 1, q, d_G.p, 1, q));
                    }
                    deflated = nd > 0;
                    chol_wa = nd == 0;
                    dev_done = true;
                    r2 = kept;
                    if (std::getenv("NLE_TRACE"))
                        fprintf(stderr, "[nle trace] Wa: %d of %d eigenvalues >= 1e-10 (largest %.3e, smallest kept %.3e), %d deflated (device)\n",
                                kept, q, esw.D[0], esw.D[kept - 1], nd);
                }
            }
        }
        if (dev_done) {
            d_L.alloc(qq);
            HIP_OK(hipMemcpyAsync(d_L.p, chw.L.p, qq * sizeof(double), hipMemcpyDeviceToDevice, sa));
        }
        HIP_OK(hipEventRecord(c->aux_ev, sa));
        HIP_OK(hipStreamWaitEvent(st, c->aux_ev, 0));  // the first stream's later kernels see F, L, G, Wa
        if (wa_serial) enqueue_gram();
    } else {
        enqueue_gram();
    }
    if (!dev_done) {
        // (see ortho_ss_prepare for why any root of the pseudo-inverse serves and when Cholesky is admissible)
        build_Wa_host();
        nd = 0;
        // A Cholesky attempt first only for small q: its small pivots come last, so on a matrix that does have eigenvalues
        // below the cut -- ten of the reference's eleven README runs, every benchmark config -- it costs most of a
        // factorisation before it proves futile (0.24 ms at q = 200).  From q = 64 on the eigenvalues come first and decide,
        // as on the device route; none below the cut is then the deflated route with nothing to deflate, i.e. the same
        // Cholesky factor.
        const bool attempt_chol = q < 64 || std::getenv("NLE_NO_DEFLATE") != nullptr || q >= 512;
        if (!dev_wa && !force_eig && attempt_chol && inv_diag <= kCholMaxInvTrace) {
            L.resize(qq);
            Li.resize(qq);
            double inv_trace = 0.0;
            chol_wa = nleh::cholesky_with_inverse(o.Wa.data(), q, L.data(), Li.data(), &inv_trace, kCholMaxInvTrace) && inv_trace <= kCholMaxInvTrace;
        }
        // Few eigenvalues below the cut (the usual case on large images: 4 of 200 at cfg4): deflate them and take the Cholesky
        // route after all.  With Vd, Ld the dropped eigenpairs and s = lambda_max, Ahat = A + Vd (s I - Ld) Vd^T has A's kept
        // eigenpairs and s on span(Vd); Ahat = L L^T, and with Pk = I - Vd Vd^T (which commutes with Ahat)
        //     F = Pk L^-T  satisfies  F F^T = Pk Ahat^-1 Pk = pinv of the kept part of A,    F^T A^2 F = L^T Pk L
        // -- the two things the device half needs.  Only the d dropped eigenvectors are formed (inverse iteration), not all q:
        // reduction + QL values + Cholesky with inverse, ~1.0 ms at q = 200 against 1.5 ms for the full eigensolve.
        // (on the host tried only where it pays: below q = 512, where the eigensolver is single threaded -- at q = 900 with 100
        // dropped eigenvalues it lost 40 ms to the threaded full solve -- and for at most q / 8 dropped eigenvalues)
        if (!dev_wa && !chol_wa && std::getenv("NLE_FORCE_EIG") == nullptr && std::getenv("NLE_NO_DEFLATE") == nullptr && q >= 16 && q < 512) {
            std::vector<double> Dbelow(max_defl + 1), Vd((size_t)q * (max_defl + 1));
            int kept = 0;
            double lam_max = 0.0, lam_min_kept = 0.0;
            tr.mark(attempt_chol ? "ss:   Wa built, Cholesky attempt" : "ss:   Wa built");
            if (nleh::sym_eigen_below(o.Wa.data(), q, NLE_EPS, max_defl, &kept, &lam_max, &lam_min_kept, Dbelow.data(), Vd.data())) {
                tr.mark("ss:   Wa eigenvalues + dropped eigenvectors");
                nd = q - kept;
                if (kept > 0 && nd <= max_defl && lam_max > 0.0) {
                    const double sig = lam_max;
                    std::vector<double> Ah(qq);
                    for (int cidx = 0; cidx < q; ++cidx)  // the symmetric matrix the reference's solver sees: lower triangle
                        for (int ridx = 0; ridx < q; ++ridx)
                            Ah[(size_t)cidx * q + ridx] = ridx >= cidx ? o.Wa[(size_t)cidx * q + ridx] : o.Wa[(size_t)ridx * q + cidx];
                    for (int t = 0; t < nd; ++t) {
                        const double wgt = sig - Dbelow[t];
                        const double* v = Vd.data() + (size_t)t * q;
                        for (int cidx = 0; cidx < q; ++cidx) {
                            const double vc = wgt * v[cidx];
                            for (int ridx = 0; ridx < q; ++ridx) Ah[(size_t)cidx * q + ridx] += v[ridx] * vc;
                        }
                    }
                    L.resize(qq);
                    Li.resize(qq);
                    double inv_trace = 0.0;
                    if (nleh::cholesky_with_inverse(Ah.data(), q, L.data(), Li.data(), &inv_trace)) {
                        tr.mark("ss:   deflated matrix + its Cholesky factor and inverse");
                        // F = L^-T - Vd (Vd^T L^-T),  G = L^T Vd
                        Fdefl.resize(qq);
                        for (int k = 0; k < q; ++k)
                            for (int a = 0; a < q; ++a) Fdefl[(size_t)k * q + a] = Li[(size_t)a * q + k];
                        Gdefl.assign((size_t)q * std::max(nd, 1), 0.0);
                        std::vector<double> wv(q);
                        for (int t = 0; t < nd; ++t) {
                            const double* v = Vd.data() + (size_t)t * q;
                            for (int k = 0; k < q; ++k) {  // w = (Vd^T L^-T)[t, k] = sum_a v[a] L^-T(a, k) = sum_a v[a] Li(k, a)
                                double acc = 0.0, g = 0.0;
                                for (int a = 0; a < q; ++a) {
                                    acc += v[a] * Li[(size_t)a * q + k];
                                    g += L[(size_t)k * q + a] * v[a];  // (L^T v)[k] = sum_a L(a, k) v[a]
                                }
                                wv[k] = acc;
                                Gdefl[(size_t)t * q + k] = g;
                            }
                            for (int k = 0; k < q; ++k)
                                for (int a = 0; a < q; ++a) Fdefl[(size_t)k * q + a] -= v[a] * wv[k];
                        }
                        deflated = nd > 0;
                        chol_wa = nd == 0;  // nothing below the cut: F = L^-T of Wa itself
                        r2 = kept;
                        if (std::getenv("NLE_TRACE"))
                            fprintf(stderr, "[nle trace] Wa: %d of %d eigenvalues >= 1e-10 (largest %.3e, smallest kept %.3e), %d deflated\n",
                                    kept, q, lam_max, lam_min_kept, nd);
                    }
                }
            }
        }
        if (!chol_wa && !deflated) {
            std::vector<double> Uf(qq), l2(q);
            if (!nleh::eigen_decomposition(o.Wa.data(), q, NLE_EPS, Uf.data(), l2.data(), &r2))
                throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge on Wa"};
            U2.assign(Uf.begin(), Uf.begin() + (size_t)q * std::max(r2, 1));
            l2_kept.assign(l2.begin(), l2.begin() + r2);
            Us.resize((size_t)q * std::max(r2, 1));
            for (int k = 0; k < r2; ++k) {
                const double sv = std::sqrt(recip0(l2[k]));
                for (int i = 0; i < q; ++i) Us[(size_t)k * q + i] = U2[(size_t)k * q + i] * sv;
            }
            if (std::getenv("NLE_TRACE"))
                fprintf(stderr, "[nle trace] Wa: %d of %d eigenvalues >= 1e-10 (largest %.3e, smallest kept %.3e)\n", r2, q, l2[0],
                        r2 > 0 ? l2[r2 - 1] : 0.0);
        }
    }
    o.r_wa = chol_wa ? q : r2;
    o.chol_wa = chol_wa;
    const bool chol_form = chol_wa || deflated;  // F is q x q and F^T A^2 F = L^T L (- G G^T)
    *host_overlapped_ms += now_ms() - h0;
    tr.mark(dev_done ? "ss: Wa root (device, second stream, beside the Gram kernels)" : "ss: Wa root (host, under the Gram kernels)");
    // ---- device: with a factor F of the (pseudo-)inverse of A = sym-lower(Wa), F F^T = A^+, the matrix the reference
    // diagonalises, Q = Wa + S (Wab Wab^T) S with S = A^+1/2 (:296), is similar on range(A) to
    //     Qt = F^T A^2 F + F^T WW F          (m x m, m = number of eigenvalues of Wa kept by the cut, :287)
    // and T2 = S Vq Sq^-1/2 = F Vt Sq^-1/2 for Qt's eigenvectors Vt (:324-327).  Eigensolver form: F = U2 L2^-1/2,
    // F^T A^2 F = L2 (diagonal); Cholesky form (no eigenvalue cut): F = L^-T, F^T A^2 F = L^T L.  On the subspace the
    // cut removed, Q acts as Wa alone -- eigenvalues < 1e-10, cut again at :313 -- so nothing is lost, and the rounding
    // of the S (..) S products (entries of S reach 1e5) can no longer lift one of them back over the cut.
    const int m = chol_form ? q : std::max(r2, 0);
    if (m <= 0) throw Fail{NLE_ERR_NUMERIC, "Wa has no eigenvalue >= 1e-10"};
    const size_t mm_ = (size_t)m * m;
    DevBuf<double> d_T(pp), d_T1((size_t)m * q), d_Qm(mm_);
    if (!dev_wa) HIP_OK(hipMemcpyAsync(d_Wa.p, o.Wa.data(), qq * sizeof(double), hipMemcpyHostToDevice, st));
    if (!dev_done) {
        std::vector<double> F;  // q x m column-major
        if (chol_form) {
            if (deflated) {
                F = Fdefl;
            } else {
                F.resize(qq);
                for (int k = 0; k < q; ++k)
                    for (int a = 0; a < q; ++a) F[(size_t)k * q + a] = Li[(size_t)a * q + k];  // L^-T
            }
            d_L.alloc(qq);
            HIP_OK(hipMemcpyAsync(d_L.p, L.data(), qq * sizeof(double), hipMemcpyHostToDevice, st));
        } else {
            F = Us;
        }
        d_F.alloc((size_t)q * m);
        HIP_OK(hipMemcpyAsync(d_F.p, F.data(), (size_t)q * m * sizeof(double), hipMemcpyHostToDevice, st));
        if (deflated && nd > 0) {
            d_G.alloc((size_t)q * nd);
            HIP_OK(hipMemcpyAsync(d_G.p, Gdefl.data(), (size_t)q * nd * sizeof(double), hipMemcpyHostToDevice, st));
        }
        HIP_OK(hipStreamSynchronize(st));  // the staging vectors go out of scope
    }
    reduce_gram();
    if (q < p) {  // samples that fall in the B block: Gk += Kr[:, q:] diag(cA[q:]^2) Kr[:, q:]^T
        std::vector<double> c2(p - q);
        for (int a = q; a < p; ++a) c2[a - q] = o.cA[a] * o.cA[a];
        DevBuf<double> d_c2(p - q);
        HIP_OK(hipMemcpyAsync(d_c2.p, c2.data(), (p - q) * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_OK(nlek::gemm64s(st, p, p, p - q, d_Kr.p + (size_t)q * p, 1, p, d_Kr.p + (size_t)q * p, p, 1, d_Gk, 1, p, nullptr,
                             d_c2.p, nullptr, d_Gk, 1, p));
        HIP_OK(hipStreamSynchronize(st));  // c2 (host) is consumed
    }
    if (r < p) {  // Gk' = P Gk P
        HIP_OK(nlek::gemm64s(st, p, p, p, d_P.p, 1, p, d_Gk, 1, p, d_T.p, 1, p));
        HIP_OK(nlek::gemm64s(st, p, p, p, d_T.p, 1, p, d_P.p, 1, p, d_Gk, 1, p));
    }
    // T1 = F^T diag(rA) Gk'[:q,:q]  (m x q);   Qt = T1 diag(rA) F (+ L^T L in the Cholesky form; + diag(l2) on the host)
    HIP_OK(nlek::gemm64s(st, m, q, q, d_F.p, q, 1, d_Gk, 1, p, d_T1.p, 1, m, nullptr, d_rA.p));
    if (chol_form) {
        DevBuf<double> d_A2(qq);
        HIP_OK(nlek::gemm64s(st, q, q, q, d_L.p, q, 1, d_L.p, 1, q, d_A2.p, 1, q));
        if (deflated && nd > 0) {  // - G G^T
            DevBuf<double> d_neg(nd);
            HIP_OK(nlek::fill64(st, d_neg.p, nd, -1.0));
            HIP_OK(nlek::gemm64s(st, q, q, nd, d_G.p, 1, q, d_G.p, q, 1, d_A2.p, 1, q, nullptr, d_neg.p, nullptr, d_A2.p, 1, q));
        }
        HIP_OK(nlek::gemm64s(st, m, m, q, d_T1.p, 1, m, d_F.p, 1, q, d_Qm.p, 1, m, nullptr, d_rA.p, nullptr, d_A2.p, 1, q));
    } else {
        HIP_OK(nlek::gemm64s(st, m, m, q, d_T1.p, 1, m, d_F.p, 1, q, d_Qm.p, 1, m, nullptr, d_rA.p));
    }
    std::vector<double> Vq, Sq;
    int rq = 0;
    // Top eigenpairs of Qt.  From dev_solver_min_n() on: reduction, eigenvalues and back-transformation on the device
    // (dense64.hip), only the inverse iteration for the K kept vectors on the host.  Below it: on the host
    // (NLE_DEVICE_TRIDIAG=1, opt-in, K <= m / 2, m <= 224: the one-workgroup reduction of tridiag.hip -- measured at m = 196:
    // 0.66 ms on the device against 0.43 ms of the 1.26 ms host solve, so it is not the default).
    const int kq = std::min(std::max(n_eig, 1), m);
    bool dev_eig = c->topk_solver == 0 && use_dev_solver(m) && !std::getenv("NLE_HOST_Q");
    DevSymEig es;  // (its staging buffer must outlive the upload it enqueues: function scope)
    DevBuf<double> d_Vq, d_l2q;
    if (dev_eig) {
        const double* d_add = nullptr;
        if (!chol_form) {
            d_l2q.alloc(m);
            HIP_OK(hipMemcpyAsync(d_l2q.p, l2_kept.data(), m * sizeof(double), hipMemcpyHostToDevice, st));
            d_add = d_l2q.p;
        }
        dev_eig = es.reduce(c, m, d_Qm.p, d_add);
        if (c->world > 1) dev_eig = ranks_where(c, !dev_eig) == 0;  // one rank on the host solver: all of them (see Wa)
    }
    const bool dev_tridiag = !dev_eig && c->topk_solver == 0 && m >= 16 && m <= nlek::tridiag_max_n() && 2 * kq <= m &&
                             std::getenv("NLE_DEVICE_TRIDIAG") != nullptr;
    if (dev_eig) {
        tr.mark("ss: Q, its tridiagonal form and eigenvalues (device)");
        h0 = now_ms();
        Sq = es.D;
        while (rq < m && Sq[rq] >= NLE_EPS) ++rq;  // :213-216
        const int Kd = std::min(n_eig, rq);
        if (Kd > 0) {
            d_Vq.alloc((size_t)m * Kd);
            es.vectors(c, 0, Kd, d_Vq.p);
        }
    } else if (dev_tridiag) {
        DevBuf<double> d_tv(mm_), d_td((size_t)3 * m), d_l2;
        const double* d_add = nullptr;
        if (!chol_form) {
            d_l2.alloc(m);
            HIP_OK(hipMemcpyAsync(d_l2.p, l2_kept.data(), m * sizeof(double), hipMemcpyHostToDevice, st));
            d_add = d_l2.p;
        }
        HIP_OK(nlek::tridiag(st, m, d_Qm.p, d_add, d_tv.p, d_td.p, d_td.p + m, d_td.p + 2 * m));
        std::vector<double> tv(mm_), td((size_t)3 * m);
        HIP_OK(hipMemcpyAsync(td.data(), d_td.p, td.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(tv.data(), d_tv.p, mm_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        tr.mark("ss: Q + its tridiagonal form on the device, download");
        h0 = now_ms();
        Vq.assign((size_t)m * kq, 0.0);
        Sq.assign(m, 0.0);
        if (!nleh::eigen_decomposition_top_reduced(m, NLE_EPS, kq, tv.data(), td.data(), td.data() + m, td.data() + 2 * m,
                                                   Vq.data(), Sq.data(), &rq))
            throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge on Q"};
    } else {
        std::vector<double> Qm(mm_);
        HIP_OK(hipMemcpyAsync(Qm.data(), d_Qm.p, mm_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        tr.mark("ss: Q on the device + download");
        // ---- host: top eigenpairs of Qt
        h0 = now_ms();
        if (!chol_form)
            for (int k = 0; k < m; ++k) Qm[(size_t)k * m + k] += l2_kept[k];
        top_eigenpairs(Qm, m, n_eig, c->topk_solver, &Vq, &Sq, &rq);
    }
    const int K = std::min(n_eig, rq);  // :314
    if (K <= 0) throw Fail{NLE_ERR_NUMERIC, "Q has no eigenvalue >= 1e-10"};
    o.K = K;
    o.r_q = rq;
    o.Sq.assign(Sq.begin(), Sq.begin() + K);
    std::vector<double> sv(K);
    for (int k = 0; k < K; ++k) sv[k] = std::sqrt(recip0(Sq[k]));  // :319-321
    *host_ms += now_ms() - h0;
    tr.mark(dev_eig ? "ss: eigenvectors of Q (inverse iteration on the host, back-transformation enqueued)" : "ss: eig(Q) (host)");
    // ---- device: T2 = F Vt Sq^-1/2, D = P[:, :q] diag(rA) T2, Vrows = [Wa T2; diag(cA_B) Kr_B D]
    DevBuf<double> d_sv(K), d_T2((size_t)q * K), d_D((size_t)p * K), d_Vr((size_t)p * K);
    if (!dev_eig) {
        d_Vq.alloc((size_t)m * K);
        HIP_OK(hipMemcpyAsync(d_Vq.p, Vq.data(), (size_t)m * K * sizeof(double), hipMemcpyHostToDevice, st));
    }
    HIP_OK(hipMemcpyAsync(d_sv.p, sv.data(), K * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(nlek::gemm64s(st, q, K, m, d_F.p, 1, q, d_Vq.p, 1, m, d_T2.p, 1, q, nullptr, nullptr, d_sv.p));
    if (r < p) {
        HIP_OK(nlek::gemm64s(st, p, K, q, d_P.p, 1, p, d_T2.p, 1, q, d_D.p, 1, p, nullptr, d_rA.p));  // first q columns of P
    } else {  // P = I, q == p: D = diag(rA) T2
        HIP_OK(hipMemcpyAsync(d_D.p, d_T2.p, (size_t)q * K * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_OK(nlek::scale_rows64(st, d_D.p, p, K, d_rA.p));
    }
    HIP_OK(nlek::gemm64s(st, q, K, q, d_Wa.p, 1, q, d_T2.p, 1, q, d_Vr.p, 1, p));  // top block of :327
    if (q < p)
        HIP_OK(nlek::gemm64s(st, p - q, K, p, d_Kr.p + q, 1, p, d_D.p, 1, p, d_Vr.p + q, 1, p, d_cA.p + q));
    o.D.resize((size_t)p * K);
    o.Vrows.resize((size_t)p * K);
    HIP_OK(hipMemcpyAsync(o.D.data(), d_D.p, o.D.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(o.Vrows.data(), d_Vr.p, o.Vrows.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    tr.mark("ss: D, Vrows on the device");
}

}  // namespace nlep
