// Host orchestration of the hot path behind the C ABI (include/nle.h): context, the
// fused train pipeline (NLEFilter::trainFilter, reference src/filter.cpp:480-502) and
// apply (:445-458).  Small (p x p, r x r) algebra and the three symmetric eigensolves run
// on the host in fp64; everything N-sized is a HIP kernel (kernels.hip).
#include "ortho.h"

using nlek::GridSpec;
using namespace nlep;

namespace {
// ---- sample set + Ka (host) ----
struct SampleSet {
    GridSpec gs;
    int p = 0;
    std::vector<long long> pix;     // row-major pixel index of each sample (permuted order)
    std::vector<float> val;         // luminance
    std::vector<float4> packed;     // {row, col, lum, 0}
    bool quantised = false;         // whole plane integer valued in [0, 255] (checked on request)
    unsigned level_tiles = 0xffffu; // then: which 16-level tiles occur in this rank's part of the plane (bit t)
};

// d_lum: base of the FULL plane -- real, or virtual when the ctx takes slab input (only rows [row0, row1) of this rank
// exist; the p sample values and the "integer valued" verdict are then completed by an all-reduce)
SampleSet fetch_samples(nle_ctx* c, const float* d_lum, const GridSpec& gs, bool check_quantised = false,
                        bool slab_plane = false) {
    SampleSet s;
    s.gs = gs;
    s.p = gs.p();
    s.val.resize(s.p);
    int flag = 1;
    int fl2[2] = {1, 0xffff};  // check_levels: [0] verdict, [1] level tiles
    if (slab_plane) {
        int row0, row1;
        slab(gs.H, c->rank, c->world, &row0, &row1);
        DevBuf<double> d_v((size_t)s.p + 1);
        PROFILED(c, NLE_K_SMALL, nlek::gather_samples_slab(c->stream, d_lum, gs, row0, row1, d_v.p));
        if (check_quantised) {
            DevBuf<int> d_flag(2);
            PROFILED(c, NLE_K_SMALL, nlek::check_levels(c->stream, d_lum + (size_t)row0 * gs.W, (long long)(row1 - row0) * gs.W, d_flag.p));
            HIP_OK(hipMemcpyAsync(fl2, d_flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
            flag = fl2[0];
        }
        const double fl = flag != 0 ? 1.0 : 0.0;
        HIP_OK(hipMemcpyAsync(d_v.p + s.p, &fl, sizeof(double), hipMemcpyHostToDevice, c->stream));
        all_reduce(c, d_v.p, (size_t)s.p + 1);
        std::vector<double> v((size_t)s.p + 1);
        HIP_OK(hipMemcpyAsync(v.data(), d_v.p, v.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < s.p; ++k) s.val[k] = (float)v[k];
        flag = v[s.p] > 0.0 ? 1 : 0;
    } else {
        DevBuf<float> d_val(s.p);
        DevBuf<int> d_flag(2);
        PROFILED(c, NLE_K_SMALL, nlek::gather_samples(c->stream, d_lum, gs, d_val.p));
        if (check_quantised) {
            PROFILED(c, NLE_K_SMALL, nlek::check_levels(c->stream, d_lum, (long long)gs.H * gs.W, d_flag.p));
            HIP_OK(hipMemcpyAsync(fl2, d_flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        }
        HIP_OK(hipMemcpyAsync(s.val.data(), d_val.p, s.p * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        if (check_quantised) flag = fl2[0];
    }
    s.quantised = check_quantised && flag == 0;
    if (s.quantised && (fl2[1] & 0xffff) != 0) s.level_tiles = (unsigned)fl2[1] & 0xffffu;
    s.pix.resize(s.p);
    s.packed.resize(s.p);
    for (int k = 0; k < s.p; ++k) {
        const int r = gs.rowOff + (k / gs.nSelCols) * gs.rowStep;
        const int cc = gs.colOff + (k % gs.nSelCols) * gs.colStep;
        s.pix[k] = (long long)r * gs.W + cc;
        s.packed[k] = make_float4((float)r, (float)cc, s.val[k], 0.f);
    }
    return s;
}

// Ka(i,j), reference src/filter.cpp:128-137,144 (fp64, integer spatial term)
std::vector<double> build_Ka(const SampleSet& s, double hx, double hy) {
    const int p = s.p;
    const double sw = 1.0 / (hx * hx), pw = 1.0 / (hy * hy);
    std::vector<double> Ka((size_t)p * p);
    auto column = [&](int j) {
        const int rj = (int)(s.pix[j] / s.gs.W), cj = (int)(s.pix[j] % s.gs.W);
        for (int i = j; i < p; ++i) {
            const int ri = (int)(s.pix[i] / s.gs.W), ci = (int)(s.pix[i] % s.gs.W);
            const long long dr = ri - rj, dc = ci - cj;
            const double sq = (double)(dr * dr + dc * dc);
            const double dv = (double)s.val[i] - (double)s.val[j];
            const double v = std::exp(-sw * sq - pw * (dv * dv));
            Ka[(size_t)j * p + i] = v;
            Ka[(size_t)i * p + j] = v;
        }
    };
    // p (p + 1) / 2 libm exponentials: 1.5 ms on one core at p = 900, before anything else of the train can start.  From
    // 384 samples on, 8 short-lived threads take BLOCKS of columns of equal triangle area (same values: every entry has one
    // writer; dealing the columns round-robin had the threads' mirrored writes share every cache line: 6.7 ms)
    const int nt = p >= 384 ? 8 : 1;
    std::vector<int> cut(nt + 1, p);
    for (int t = 0; t < nt; ++t) cut[t] = (int)(p * (1.0 - std::sqrt(1.0 - (double)t / nt)));
    nleh::run_parts(nt, nt, [&](int t) {
        for (int j = cut[t]; j < cut[t + 1]; ++j) column(j);
    });
    return Ka;
}

// What the later stages need of Ka.  The literal form is the reference's: Ka's eigenpairs with the
// cut at 1e-10 (VA = V_r, lam, B = V_r / lam).  The Phi-free path uses them only through
//   VA diag(lam) VA^T (= Ka restricted to its range),  B diag(lam) VA^T (= the projector P on that range)
//   and B diag(lam) B^T (= pinv(Ka)),
// so when Ka is provably full rank at the reference's threshold (every eigenvalue >= 1e-10, certified
// by 1 / trace(Ka^-1)) the Cholesky factor serves as well: VA = L, lam = 1, B = L^-T (P = I) -- a
// p^3/3 factorisation instead of a p x p eigensolve.  The materialised path keeps the eigenpairs.


Nystrom solve_Ka(nle_ctx* c, const std::vector<double>& Ka, int p, bool allow_chol) {
    Nystrom n;
    if (allow_chol && std::getenv("NLE_FORCE_EIG") == nullptr) {
        if (c && use_dev_solver(p) && !std::getenv("NLE_HOST_KA")) {  // blocked factorisation + inverse on the device (dense64.hip)
            const size_t pp = (size_t)p * p;
            auto kd = std::make_shared<KaDevice>();
            kd->Ka.alloc(pp);
            upload_staged(c, kd->Ka.p, Ka.data(), pp, c->stream);
            kd->ch.factor(c, p, kd->Ka.p);
            if (kd->ch.finish(c) && kd->ch.inv_trace <= kCholMaxInvTrace) {
                // the factors stay on the device: the table path builds the Sinkhorn update's operands from them there
                // (22 ms of host factorisation and ~28 ms of host transposes and uploads at p = 900 become ~6 ms)
                n.chol = true;
                n.r = p;
                n.ldr = ld4(p);
                n.lam.assign(p, 1.0);
                n.Ka = Ka;
                n.dev = std::move(kd);
                return n;
            }
        } else {
            std::vector<double> L((size_t)p * p), Li((size_t)p * p);
            double inv_trace = 0.0;
            if (nleh::cholesky_with_inverse(Ka.data(), p, L.data(), Li.data(), &inv_trace, kCholMaxInvTrace) &&
                inv_trace <= kCholMaxInvTrace) {
                n.chol = true;
                n.r = p;
                n.ldr = ld4(p);
                n.VA = std::move(L);
                n.lam.assign(p, 1.0);
                n.B.resize((size_t)p * p);
                for (int k = 0; k < p; ++k)
                    for (int a = 0; a < p; ++a) n.B[(size_t)k * p + a] = Li[(size_t)a * p + k];  // L^-T
                n.Ka = Ka;
                return n;
            }
        }
    }
    // nystromApproximation, reference src/filter.cpp:262-271
    std::vector<double> U((size_t)p * p), D(p);
    int r = 0;
    if (!nleh::eigen_decomposition(Ka.data(), p, NLE_EPS, U.data(), D.data(), &r))
        throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge on Ka"};
    int nnz = 0;
    for (int k = 0; k < r; ++k)
        if (std::fabs(D[k]) >= NLE_EPS) ++nnz;  // inplaceReciprocal count (:266)
    r = std::min(r, nnz);
    if (r <= 0) throw Fail{NLE_ERR_NUMERIC, "Ka has no eigenvalue >= 1e-10"};
    n.r = r;
    n.ldr = ld4(r);
    n.VA.assign(U.begin(), U.begin() + (size_t)p * r);
    n.lam.assign(D.begin(), D.begin() + r);
    n.B.resize((size_t)p * r);
    for (int k = 0; k < r; ++k) {
        const double inv = recip0(n.lam[k]);  // :265-268
        for (int a = 0; a < p; ++a) n.B[(size_t)k * p + a] = n.VA[(size_t)k * p + a] * inv;
    }
    return n;
}

// B = V_A diag(1/lambda) as fp32 row-major p x ldr
std::vector<float> build_B(const Nystrom& n, int p) {
    std::vector<float> B((size_t)p * n.ldr, 0.f);
    for (int k = 0; k < n.r; ++k) {
        const double inv = recip0(n.lam[k]);
        for (int s = 0; s < p; ++s) B[(size_t)s * n.ldr + k] = (float)(n.VA[(size_t)k * p + s] * inv);
    }
    return B;
}

// Phi for the local slab: fused affinity + Nystrom extension, then exact V_A sample rows
void build_phi(nle_ctx* c, const float* d_lum, const SampleSet& ss, const Nystrom& ny, double hx,
               double hy, long long pix0, long long M, float* d_phi) {
    const int p = ss.p;
    DevBuf<float4> d_samples(p);
    HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), p * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    std::vector<float> B = build_B(ny, p);
    DevBuf<float> d_B(B.size());
    HIP_OK(hipMemcpyAsync(d_B.p, B.data(), B.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const float sw = nsw_of(hx), pw = nsw_of(hy);
    static const bool env3 = std::getenv("NLE_NYSTROM_BF16X3") != nullptr;
    if (c->nystrom_bf16x3 || env3) {  // split-bf16 operands on the bf16 matrix cores (tsgemm_bf16x3.hip)
        DevBuf<unsigned short> d_Bs(nlek::ts_gemm_bf16x3_bsplit_elems(p, ny.ldr));
        HIP_OK(nlek::ts_gemm_bf16x3_split(c->stream, d_B.p, p, ny.ldr, d_Bs.p));
        PROFILED(c, NLE_K_NYSTROM, nlek::ts_gemm_bf16x3(c->stream, d_lum, ss.gs, d_samples.p, sw, pw, pix0, d_Bs.p, ny.ldr, p,
                                                        d_phi, ny.ldr, M));
    } else {
        PROFILED(c, NLE_K_NYSTROM, nlek::ts_gemm(c->stream, true, nullptr, 0, d_lum, ss.gs, d_samples.p, sw, pw, pix0,
                                                 d_B.p, ny.ldr, p, d_phi, ny.ldr, M, nullptr, NLE_EPS));
    }
    // sample pixels carry their exact V_A row (top block of phi, reference :275)
    std::vector<float> rows;
    std::vector<long long> idx;
    for (int k = 0; k < p; ++k) {
        const long long loc = ss.pix[k] - pix0;
        if (loc < 0 || loc >= M) continue;
        idx.push_back(loc);
        const size_t off = rows.size();
        rows.resize(off + ny.ldr, 0.f);
        for (int j = 0; j < ny.r; ++j) rows[off + j] = (float)ny.VA[(size_t)j * p + k];
    }
    DevBuf<float> d_rows(rows.size());
    DevBuf<long long> d_idx(idx.size());
    if (!idx.empty()) {
        HIP_OK(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_idx.p, idx.data(), idx.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
        PROFILED(c, NLE_K_SMALL, nlek::scatter_rows(c->stream, d_rows.p, d_idx.p, (int)idx.size(), ny.ldr, d_phi, M));
    }
    HIP_OK(hipStreamSynchronize(c->stream));  // host staging vectors go out of scope
}

// Sinkhorn (reference :238-245) as 2T passes: t0 = Phi^T 1, then alternately
// t <- Phi^T recip(Phi (lam o t)).  Returns u_c, u_r (host) and leaves lam o t_c_in on d_u_c.
inline hipError_t rowpass_any(hipStream_t s, int mode, const float* X, long long M, int ld, const double* t, const double* lam,
                              const float* xv, double eps, double* partial, int* nb) {
    return nlek::rowpass(s, mode, X, M, ld, t, lam, xv, eps, partial, nb);
}
inline hipError_t rowpass_any(hipStream_t s, int mode, const double* X, long long M, int ld, const double* t, const double* lam,
                              const float* xv, double eps, double* partial, int* nb) {
    return nlek::rowpass64(s, mode, X, M, ld, t, lam, xv, eps, partial, nb);
}

template <typename T_>
void sinkhorn_passes(nle_ctx* c, const T_* d_phi, long long M, int ld, int r,
                     const std::vector<double>& lam, int T, std::vector<double>* u_c,
                     std::vector<double>* u_r, double* d_u_c_out /* ld doubles or null */) {
    if (T < 1) throw Fail{NLE_ERR_INVALID, "nSinkhornIter must be >= 1"};
    std::vector<double> lam_pad(ld, 0.0);
    std::copy(lam.begin(), lam.begin() + r, lam_pad.begin());
    DevBuf<double> d_lam(ld), d_t[3], d_partial((size_t)nlek::kRowpassMaxBlocks * ld);
    for (auto& b : d_t) b.alloc(ld);
    HIP_OK(hipMemcpyAsync(d_lam.p, lam_pad.data(), ld * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int nb = 0;
    // t_r(0) = Phi^T 1
    PROFILED(c, NLE_K_SINKHORN_PASS, rowpass_any(c->stream, nlek::ROWPASS_COLSUM, d_phi, M, ld, nullptr, nullptr,
                                                   nullptr, NLE_EPS, d_partial.p, &nb));
    PROFILED(c, NLE_K_REDUCE, nlek::reduce_partials(c->stream, d_partial.p, nb, ld, d_t[0].p));
    all_reduce(c, d_t[0].p, ld);
    // cur = index of the t feeding the next pass
    int cur = 0;
    int idx_c_in = 0, idx_r_in = 0;
    for (int it = 0; it < T; ++it) {
        // c = recip(Phi (lam o t_r));  t_c = Phi^T c
        idx_c_in = cur;
        int nxt = (cur + 1) % 3;
        PROFILED(c, NLE_K_SINKHORN_PASS, rowpass_any(c->stream, nlek::ROWPASS_RECIP, d_phi, M, ld, d_t[cur].p,
                                                       d_lam.p, nullptr, NLE_EPS, d_partial.p, &nb));
        PROFILED(c, NLE_K_REDUCE, nlek::reduce_partials(c->stream, d_partial.p, nb, ld, d_t[nxt].p));
        all_reduce(c, d_t[nxt].p, ld);
        cur = nxt;
        idx_r_in = cur;
        if (it + 1 < T) {
            // r = recip(Phi (lam o t_c));  t_r = Phi^T r  (not needed after the last iteration:
            // only u_r = lam o t_c enters the W blocks)
            nxt = (cur + 1) % 3;
            if (nxt == idx_c_in) nxt = (nxt + 1) % 3;
            PROFILED(c, NLE_K_SINKHORN_PASS, rowpass_any(c->stream, nlek::ROWPASS_RECIP, d_phi, M, ld, d_t[cur].p,
                                                           d_lam.p, nullptr, NLE_EPS, d_partial.p, &nb));
            PROFILED(c, NLE_K_REDUCE, nlek::reduce_partials(c->stream, d_partial.p, nb, ld, d_t[nxt].p));
            all_reduce(c, d_t[nxt].p, ld);
            cur = nxt;
        }
    }
    std::vector<double> tc(ld), tr(ld);
    HIP_OK(hipMemcpyAsync(tc.data(), d_t[idx_c_in].p, ld * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(tr.data(), d_t[idx_r_in].p, ld * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (d_u_c_out) PROFILED(c, NLE_K_SMALL, nlek::scale_vec(c->stream, d_lam.p, d_t[idx_c_in].p, ld, d_u_c_out));
    HIP_OK(hipStreamSynchronize(c->stream));
    u_c->assign(r, 0.0);
    u_r->assign(r, 0.0);
    for (int k = 0; k < r; ++k) {
        (*u_c)[k] = lam[k] * tc[k];
        (*u_r)[k] = lam[k] * tr[k];
    }
}

// G (r x r col-major) = sum_i c_i^2 phi_i phi_i^T over ALL rows of every rank
std::vector<double> gram_all(nle_ctx* c, const float* d_phi, long long M, int ld, int r, const double* d_u) {
    const int ntiles = nlek::gram_num_tiles(ld);
    DevBuf<double> d_partial(std::max<size_t>(nlek::gram_partial_elems(std::max<long long>(M, 1), ld), 1));
    DevBuf<double> d_tiles((size_t)ntiles * 1024);
    if (M > 0) {
        PROFILED(c, NLE_K_GRAM, nlek::gram(c->stream, d_phi, M, ld, d_u, NLE_EPS, d_partial.p, d_tiles.p));
    } else {
        HIP_OK(hipMemsetAsync(d_tiles.p, 0, (size_t)ntiles * 1024 * sizeof(double), c->stream));
    }
    all_reduce(c, d_tiles.p, (size_t)ntiles * 1024);
    std::vector<double> tiles((size_t)ntiles * 1024);
    HIP_OK(hipMemcpyAsync(tiles.data(), d_tiles.p, tiles.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> G((size_t)r * r, 0.0);
    const int nt = (ld + 31) / 32;
    int t = 0;
    for (int ti = 0; ti < nt; ++ti)
        for (int tj = ti; tj < nt; ++tj, ++t) {
            const double* tl = tiles.data() + (size_t)t * 1024;
            for (int a = 0; a < 32; ++a)
                for (int b = 0; b < 32; ++b) {
                    const int i = ti * 32 + a, j = tj * 32 + b;
                    if (i >= r || j >= r) continue;
                    if (ti == tj && j < i) continue;  // diagonal tiles: take the upper half
                    const double v = tl[a * 32 + b];
                    G[(size_t)j * r + i] = v;
                    G[(size_t)i * r + j] = v;
                }
        }
    return G;
}

}  // namespace

namespace {
// unpack the upper-triangular 32x32 tile list of gram()/gram_fused() into a symmetric n x n matrix
std::vector<double> unpack_tiles(const std::vector<double>& tiles, int ld, int n, int ts) {
    std::vector<double> G((size_t)n * n, 0.0);
    const int nt = (ld + ts - 1) / ts;
    int t = 0;
    for (int ti = 0; ti < nt; ++ti)
        for (int tj = ti; tj < nt; ++tj, ++t) {
            const double* tl = tiles.data() + (size_t)t * ts * ts;
            for (int a = 0; a < ts; ++a)
                for (int b = 0; b < ts; ++b) {
                    const int i = ti * ts + a, j = tj * ts + b;
                    if (i >= n || j >= n) continue;
                    if (ti == tj && j < i) continue;  // diagonal tiles: take the upper half
                    const double v = tl[a * ts + b];
                    G[(size_t)j * n + i] = v;
                    G[(size_t)i * n + j] = v;
                }
        }
    return G;
}

void scatter_sample_rows(nle_ctx* c, const std::vector<long long>& pix, int nrows, const std::vector<double>& rows_cm,
                         int ldrows, int K, int ldv, long long pix0, long long M, float* d_V) {
    std::vector<float> rows;
    std::vector<long long> idx;
    for (int a = 0; a < nrows; ++a) {
        const long long loc = pix[a] - pix0;
        if (loc < 0 || loc >= M) continue;
        idx.push_back(loc);
        const size_t off = rows.size();
        rows.resize(off + ldv, 0.f);
        for (int k = 0; k < K; ++k) rows[off + k] = (float)rows_cm[(size_t)k * ldrows + a];
    }
    if (idx.empty()) return;
    DevBuf<float> d_rows(rows.size());
    DevBuf<long long> d_idx(idx.size());
    HIP_OK(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(d_idx.p, idx.data(), idx.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    PROFILED(c, NLE_K_SMALL, nlek::scatter_rows(c->stream, d_rows.p, d_idx.p, (int)idx.size(), ldv, d_V, M));
    HIP_OK(hipStreamSynchronize(c->stream));  // staging vectors go out of scope
}

// ---- the two train paths; both fill f->K, ldv, eigvals, d_V ----
struct StageMs {
    double sinkhorn = 0, gram = 0, project = 0, host = 0, host_overlapped = 0;
};

// (1) materialised Phi: Phi = K_AB^T B written once (N x r fp32), streamed by every later pass
void train_materialised(nle_ctx* c, nle_filter* f, const float* d_lum, const SampleSet& ss, const Nystrom& ny,
                        double hx, double hy, int T, int n_eig, long long pix0, long long M, StageMs* ms) {
    Timer tm_s(c->stream), tm_g(c->stream), tm_p(c->stream);
    tm_s.start();
    DevBuf<float> d_phi((size_t)std::max<long long>(M, 1) * ny.ldr);
    build_phi(c, d_lum, ss, ny, hx, hy, pix0, M, d_phi.p);
    std::vector<double> u_c, u_r;
    DevBuf<double> d_u_c(ny.ldr);
    sinkhorn_passes(c, d_phi.p, M, ny.ldr, ny.r, ny.lam, T, &u_c, &u_r, d_u_c.p);
    tm_s.stop();
    tm_g.start();
    std::vector<double> G = gram_all(c, d_phi.p, M, ny.ldr, ny.r, d_u_c.p);
    tm_g.stop();
    double h0 = now_ms();
    Ortho o = orthogonalize_host(ny, ss.p, u_c, u_r, std::move(G), n_eig, true, c->topk_solver);
    ms->host += now_ms() - h0;
    f->K = o.K;
    f->ldv = ld4(o.K);
    f->eigvals = o.Sq;
    f->formulation = NLE_MODE_MATERIALISED;
    f->r_wa = o.r_wa;
    f->r_q = o.r_q;
    tm_p.start();
    std::vector<float> Cp((size_t)ny.r * f->ldv, 0.f);
    for (int k = 0; k < o.K; ++k)
        for (int j = 0; j < ny.r; ++j) Cp[(size_t)j * f->ldv + k] = (float)o.Cproj[(size_t)k * ny.r + j];
    DevBuf<float> d_Cp(Cp.size());
    HIP_OK(hipMemcpyAsync(d_Cp.p, Cp.data(), Cp.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    DevBuf<float> d_V((size_t)std::max<long long>(M, 1) * f->ldv);
    PROFILED(c, NLE_K_PROJECT, nlek::ts_gemm(c->stream, false, d_phi.p, ny.ldr, nullptr, ss.gs, nullptr, 0.f, 0.f, 0,
                                             d_Cp.p, f->ldv, ny.r, d_V.p, f->ldv, M, d_u_c.p, NLE_EPS));
    scatter_sample_rows(c, ss.pix, o.q, o.VArows, o.q, o.K, f->ldv, pix0, M, d_V.p);
    tm_p.stop();
    HIP_OK(hipStreamSynchronize(c->stream));
    f->v_bytes = d_V.n * sizeof(float);
    f->d_V = d_V.take();
    ms->sinkhorn = tm_s.ms();
    ms->gram = tm_g.ms();
    ms->project = tm_p.ms();
}

// (1b) the same literal decomposition with Phi and V in fp64 (generic64.hip): what auto mode falls back to when the
// table form does not apply, and what the 1e-4 bar needs on inputs whose detail layers are small differences
void build_phi64(nle_ctx* c, const float* d_lum, const SampleSet& ss, const Nystrom& ny, double hx, double hy, long long pix0,
                 long long M, double* d_phi) {
    const int p = ss.p, ldp = ld4(p), r = ny.r, ldr = ny.ldr;
    DevBuf<float4> d_samples(p);
    HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), p * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    DevBuf<double> d_B(ny.B.size());  // p x r column-major = what ts_gemm64 takes
    HIP_OK(hipMemcpyAsync(d_B.p, ny.B.data(), ny.B.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemsetAsync(d_phi, 0, (size_t)std::max<long long>(M, 1) * ldr * sizeof(double), c->stream));
    const long long chunk = 1ll << 20;  // affinity rows of 1 Mi pixels at a time (K_AB is never held whole)
    DevBuf<double> d_kab((size_t)std::min<long long>(std::max<long long>(M, 1), chunk) * ldp);
    for (long long i0 = 0; i0 < M; i0 += chunk) {
        const long long m = std::min(chunk, M - i0);
        PROFILED(c, NLE_K_AFFINITY, nlek::affinity64(c->stream, d_lum, ss.gs, d_samples.p, p, ldp, 1.0 / (hx * hx),
                                                     1.0 / (hy * hy), pix0 + i0, m, d_kab.p));
        PROFILED(c, NLE_K_NYSTROM, nlek::ts_gemm64(c->stream, d_kab.p, m, ldp, p, d_B.p, r, nullptr, d_phi + (size_t)i0 * ldr, ldr));
    }
    std::vector<double> rows;
    std::vector<long long> idx;
    for (int k = 0; k < p; ++k) {  // sample pixels carry their exact V_A row (top block of phi, reference :275)
        const long long loc = ss.pix[k] - pix0;
        if (loc < 0 || loc >= M) continue;
        idx.push_back(loc);
        const size_t off = rows.size();
        rows.resize(off + ldr, 0.0);
        for (int j = 0; j < r; ++j) rows[off + j] = ny.VA[(size_t)j * p + k];
    }
    DevBuf<double> d_rows(rows.size());
    DevBuf<long long> d_idx(idx.size());
    if (!idx.empty()) {
        HIP_OK(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_idx.p, idx.data(), idx.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
        PROFILED(c, NLE_K_SMALL, nlek::scatter_rows64(c->stream, d_rows.p, d_idx.p, (int)idx.size(), ldr, d_phi, M));
    }
    HIP_OK(hipStreamSynchronize(c->stream));  // host staging vectors go out of scope
}

// G (r x r col-major) = sum over ALL rows of every rank of c_i^2 phi_i phi_i^T, c_i = recip(phi_i . u) (d_u null: 1)
std::vector<double> gram_all64(nle_ctx* c, const double* d_phi, long long M, int ld, int r, const double* d_u) {
    DevBuf<double> d_cs, d_part(std::max<size_t>(nlek::gram64d_partial_elems(std::max<long long>(M, 1), r), 1)), d_G((size_t)r * r);
    if (d_u && M > 0) {
        d_cs.alloc((size_t)M);
        PROFILED(c, NLE_K_SMALL, nlek::row_scalings64(c->stream, d_phi, M, ld, r, d_u, NLE_EPS, d_cs.p));
    }
    PROFILED(c, NLE_K_GRAM, nlek::gram64d(c->stream, d_phi, M, ld, r, d_cs.p, d_part.p, d_G.p));
    all_reduce(c, d_G.p, (size_t)r * r);
    std::vector<double> G((size_t)r * r);
    HIP_OK(hipMemcpyAsync(G.data(), d_G.p, G.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return G;
}

void train_generic64(nle_ctx* c, nle_filter* f, const float* d_lum, const SampleSet& ss, const Nystrom& ny, double hx,
                     double hy, int T, int n_eig, long long pix0, long long M, StageMs* ms) {
    Timer tm_s(c->stream), tm_g(c->stream), tm_p(c->stream);
    tm_s.start();
    const size_t phi_elems = (size_t)std::max<long long>(M, 1) * ny.ldr;
    size_t free_b = 0, total_b = 0;
    const bool no_fit = hipMemGetInfo(&free_b, &total_b) == hipSuccess && phi_elems * sizeof(double) > free_b + c->arena_bytes;
    if (ranks_where(c, no_fit) > 0)  // refused on every rank if it does not fit on one (nobody is left in a collective)
        throw Fail{NLE_ERR_INVALID, "fp64 formulation: Phi (N x r doubles) does not fit in device memory; use an integer-valued "
                                    "luminance plane with a sample grid of at most 32 x 36 (table formulation) or NLE_MODE_MATERIALISED"};
    DevBuf<double> d_phi(phi_elems);
    build_phi64(c, d_lum, ss, ny, hx, hy, pix0, M, d_phi.p);
    std::vector<double> u_c, u_r;
    DevBuf<double> d_u_c(ny.ldr);
    sinkhorn_passes(c, d_phi.p, M, ny.ldr, ny.r, ny.lam, T, &u_c, &u_r, d_u_c.p);
    tm_s.stop();
    tm_g.start();
    std::vector<double> G = gram_all64(c, d_phi.p, M, ny.ldr, ny.r, d_u_c.p);
    tm_g.stop();
    double h0 = now_ms();
    Ortho o = orthogonalize_host(ny, ss.p, u_c, u_r, std::move(G), n_eig, /*device_f32=*/false, c->topk_solver);
    ms->host += now_ms() - h0;
    f->K = o.K;
    f->ldv = ld4(o.K);
    f->eigvals = o.Sq;
    f->formulation = NLE_MODE_MATERIALISED_F64;
    f->r_wa = o.r_wa;
    f->r_q = o.r_q;
    tm_p.start();
    DevBuf<double> d_Cp(o.Cproj.size()), d_cs((size_t)std::max<long long>(M, 1));  // Cproj: r x K column-major
    HIP_OK(hipMemcpyAsync(d_Cp.p, o.Cproj.data(), o.Cproj.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PROFILED(c, NLE_K_SMALL, nlek::row_scalings64(c->stream, d_phi.p, M, ny.ldr, ny.r, d_u_c.p, NLE_EPS, d_cs.p));
    DevBuf<double> d_V((size_t)std::max<long long>(M, 1) * f->ldv);
    HIP_OK(hipMemsetAsync(d_V.p, 0, d_V.n * sizeof(double), c->stream));
    PROFILED(c, NLE_K_PROJECT, nlek::ts_gemm64(c->stream, d_phi.p, M, ny.ldr, ny.r, d_Cp.p, o.K, d_cs.p, d_V.p, f->ldv));
    {   // exact rows of the A block (top block of :327)
        std::vector<double> rows;
        std::vector<long long> idx;
        for (int a = 0; a < o.q; ++a) {
            const long long loc = ss.pix[a] - pix0;
            if (loc < 0 || loc >= M) continue;
            idx.push_back(loc);
            const size_t off = rows.size();
            rows.resize(off + f->ldv, 0.0);
            for (int k = 0; k < o.K; ++k) rows[off + k] = o.VArows[(size_t)k * o.q + a];
        }
        if (!idx.empty()) {
            DevBuf<double> d_rows(rows.size());
            DevBuf<long long> d_idx(idx.size());
            HIP_OK(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIP_OK(hipMemcpyAsync(d_idx.p, idx.data(), idx.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
            PROFILED(c, NLE_K_SMALL, nlek::scatter_rows64(c->stream, d_rows.p, d_idx.p, (int)idx.size(), f->ldv, d_V.p, M));
            HIP_OK(hipStreamSynchronize(c->stream));
        }
    }
    tm_p.stop();
    HIP_OK(hipStreamSynchronize(c->stream));
    f->v64_bytes = d_V.n * sizeof(double);
    f->d_V64 = d_V.take();
    ms->sinkhorn = tm_s.ms();
    ms->gram = tm_g.ms();
    ms->project = tm_p.ms();
}

// Operands of the factored Sinkhorn update (fused.hip: k_sink_update_a/b) from what solve_Ka left: X1 (2p x r column-major)
// and X2 (2p x r row-major) = [B; V_A], lambda.
void build_update_operands(nle_ctx* c, const Nystrom& ny, int p, DevBuf<double>& d_X1, DevBuf<double>& d_X2,
                           DevBuf<double>& d_lam) {
    const int r = ny.r;
    if (ny.dev) {
        // Cholesky form with the factors on the device: X1 = [L^-T; 0] (2p x p column-major), X2 = [L^-T; Ka] row-major --
        // row a of L^-T is column a of L^-1 and Ka is symmetric, so X2 is two plain copies and X1 one transpose
        const size_t n2 = (size_t)2 * p, pp = (size_t)p * p;
        d_X1.alloc(n2 * p);
        d_X2.alloc(n2 * p);
        d_lam.alloc(p);
        HIP_OK(hipMemsetAsync(d_X1.p, 0, n2 * p * sizeof(double), c->stream));
        HIP_OK(nlek::transpose64(c->stream, p, ny.dev->ch.Linv.p, d_X1.p, p, 2 * p));
        HIP_OK(hipMemcpyAsync(d_X2.p, ny.dev->ch.Linv.p, pp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_X2.p + pp, ny.dev->Ka.p, pp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_OK(nlek::fill64(c->stream, d_lam.p, p, 1.0));
    } else {
        // X1 (2p x r column-major) and X2 (2p x r row-major) = [B; V_A]; Cholesky form: X1 = [L^-T; 0], the lower
        // half of X2 = the rows of Ka itself (exact projector / exact V_A diag(lambda) V_A^T, see k_sink_update_b)
        const size_t n2 = (size_t)2 * p;
        std::vector<double> X1(n2 * r, 0.0), X2(n2 * r);
        for (int k = 0; k < r; ++k)
            for (int a = 0; a < p; ++a) {
                const double b = ny.B[(size_t)k * p + a];
                const double va = ny.chol ? ny.Ka[(size_t)k * p + a] : ny.VA[(size_t)k * p + a];  // Ka symmetric
                X1[(size_t)k * n2 + a] = b;
                if (!ny.chol) X1[(size_t)k * n2 + p + a] = va;
                X2[(size_t)a * r + k] = b;
                X2[(size_t)(p + a) * r + k] = va;
            }
        d_X1.alloc(X1.size());
        d_X2.alloc(X2.size());
        d_lam.alloc(r);
        HIP_OK(hipMemcpyAsync(d_X1.p, X1.data(), X1.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_X2.p, X2.data(), X2.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_lam.p, ny.lam.data(), r * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));  // the staging vectors go out of scope (the column-sum pass is done by now)
    }
}

// (2) Phi-free: every N-sized pass regenerates its affinity rows (fused.hip)
// `solve` factors Ka on the host (solve_Ka); it is called only after the first pass -- the column sum, which
// needs nothing of it -- is on the stream, so the factorisation runs under that pass.
void train_sample_space(nle_ctx* c, nle_filter* f, const float* d_lum, const SampleSet& ss,
                        const std::function<Nystrom()>& solve, double hx, double hy, int T, int n_eig, long long pix0,
                        long long M, StageMs* ms) {
    const int p = ss.p;
    const int P64 = nlek::sink_pass_ld(p);
    const float nsw = nsw_of(hx), npw = nsw_of(hy);
    Trace tr;
    Timer tm_s(c->stream), tm_g(c->stream), tm_p(c->stream);
    tm_s.start();
    // the pass kernel reads the sample table up to the next multiple of 16: pad with zeros (their
    // w entries are zero, so they only have to be finite)
    DevBuf<float4> d_samples(P64);
    HIP_OK(hipMemsetAsync(d_samples.p, 0, P64 * sizeof(float4), c->stream));
    HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), p * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    constexpr int kZS = 8;  // slices of the block partials, summed by k_sink_update
    DevBuf<double> d_z((size_t)kZS * P64), d_w(P64), d_sAh((size_t)2 * T * p), d_X1, d_X2, d_lam, d_uv((size_t)3 * p),
        d_partial((size_t)nlek::sink_pass_rows(std::max<long long>(M, 1)) * P64);
    DevBuf<double> d_cbuf((size_t)std::max<long long>(M, 1));
    // quantised luminance + Cartesian sample grid: table look-ups replace the exponentials (fused.hip)
    const bool hist = ss.quantised && c->mode != 3 && ss.gs.nSelCols <= nlek::sink_hist_max_cols() &&
                      ss.gs.nSelRows <= 32 && M > 0;
    if (!hist && p > nlek::sink_pass_max_p()) throw Fail{NLE_ERR_INVALID, "Phi-free path: too many samples for the generic kernels"};
    const int nrows_local = (int)(M / ss.gs.W), row0 = (int)(pix0 / ss.gs.W);
    DevBuf<double> d_er, d_ecT, d_Ep;
    if (hist) {
        d_er.alloc((size_t)nrows_local * ss.gs.nSelRows);
        d_ecT.alloc((size_t)ss.gs.nSelCols * ss.gs.W);
        d_Ep.alloc((size_t)256 * p);
        d_partial.alloc((size_t)nrows_local * P64);
        PROFILED(c, NLE_K_SMALL, nlek::hist_tables(c->stream, ss.gs, d_samples.p, p, hx, hy, row0, nrows_local, d_er.p,
                                                   d_ecT.p, d_Ep.p));
    }
    HIP_OK(hipMemsetAsync(d_w.p, 0, P64 * sizeof(double), c->stream));
    const bool hist_tiled = hist && ss.gs.nSelCols <= 36 && ss.gs.nSelRows <= 32 && std::getenv("NLE_HIST_UNTILED") == nullptr;
    DevBuf<double> d_hws;
    if (hist_tiled) d_hws.alloc(nlek::hist_tiled_workspace_elems(ss.gs, nrows_local));
    // level-sorted rows: the pixel halves of every table pass run without LDS atomics (sorted.hip); sorted once here
    const bool sorted = hist_tiled && ss.gs.W <= nlek::sorted_max_width() && std::getenv("NLE_NO_SORTED_ROWS") == nullptr;
    DevBuf<unsigned short> d_scol, d_first;
    DevBuf<uint2> d_desc;
    DevBuf<double> d_E, d_E2;
    nlek::SortedRows sr{};
    if (sorted) {
        d_scol.alloc(nlek::sorted_scol_elems(ss.gs.W, nrows_local));  // k_sort_rows writes every entry a pass reads
        d_first.alloc((size_t)nrows_local * 258);
        d_desc.alloc((size_t)nrows_local * nlek::kSortedThreads);
        d_E.alloc((size_t)ss.gs.W + 1);
        PROFILED(c, NLE_K_SMALL, nlek::dist_table(c->stream, ss.gs.W, hx, d_E.p));
        PROFILED(c, NLE_K_SMALL, nlek::sort_rows(c->stream, d_lum, ss.gs, row0, nrows_local, d_scol.p, d_desc.p, d_first.p));
        HIP_OK(hipMemsetAsync(d_cbuf.p, 0, d_cbuf.n * sizeof(double), c->stream));  // sample pixels are never visited
        sr = nlek::SortedRows{d_scol.p, d_desc.p, d_first.p, d_E.p, false, 0.0};
        sr.rec = nlek::sorted_recurrence(ss.gs, hx, &sr.kappa);
        sr.mom = nlek::sorted_moments_ok(ss.gs, hx);
        if (std::getenv("NLE_ALL_LEVEL_TILES") == nullptr) {  // the tables' columns of level tiles that do not occur are skipped
            int t0 = 0, t1 = 16;
            while (t0 < 15 && !((ss.level_tiles >> t0) & 1u)) ++t0;
            while (t1 > t0 + 1 && !((ss.level_tiles >> (t1 - 1)) & 1u)) --t1;
            sr.lev_t0 = t0;
            sr.lev_nt = t1 - t0;
        }
        if (nlek::sorted_gsum_ok(ss.gs, hx)) {  // the Gram on index sums: one more distance table, exp(-2 d^2 / hx^2)
            d_E2.alloc((size_t)ss.gs.W + 1);
            PROFILED(c, NLE_K_SMALL, nlek::dist_table(c->stream, ss.gs.W, hx / std::sqrt(2.0), d_E2.p));
            sr.E2 = d_E2.p;
            sr.hx = hx;
        }
    }
    const nlek::SortedRows* srp = sorted ? &sr : nullptr;
    const int nrows = hist ? nrows_local : nlek::sink_pass_rows(std::max<long long>(M, 1));
    tr.mark("ss: alloc+upload");
    // pass n uses the scaling whose sample row sums are sAh[n-1] (and w) and produces sAh[n]; pass 0 is the
    // column sum Phi^T 1 (:234,239)
    auto pass_pixels = [&](int mode, double* ybuf) {  // the N-sized half: z = sum over this rank's pixels
        if (M > 0 && hist_tiled) {
            // tiled table pass: writes the local column sums straight into slice 0 of d_z
            static const int kmap[4] = {NLE_K_SINK_TABLES, NLE_K_SINKHORN_PASS, NLE_K_REDUCE, NLE_K_REDUCE};
            ProfObserver obs(c, kmap);
            HIP_OK(nlek::sink_hist_tiled(c->stream, mode, d_lum, ss.gs, p, P64, row0, nrows_local, d_er.p, d_ecT.p,
                                         d_Ep.p, d_w.p, NLE_EPS, ybuf, d_hws.p, d_z.p, &obs, nullptr, nullptr, srp));
        } else if (M > 0) {
            if (hist)
                PROFILED(c, NLE_K_SINKHORN_PASS, nlek::sink_hist(c->stream, mode, d_lum, ss.gs, p, P64, row0, nrows_local,
                                                                 d_er.p, d_ecT.p, d_Ep.p, d_w.p, NLE_EPS, ybuf,
                                                                 d_partial.p));
            else
                PROFILED(c, NLE_K_SINKHORN_PASS, nlek::sink_pass(c->stream, mode, d_lum, ss.gs, d_samples.p, p, d_w.p, nsw,
                                                                 npw, pix0, M, NLE_EPS, ybuf, d_partial.p));
            PROFILED(c, NLE_K_REDUCE, nlek::reduce_partials(c->stream, d_partial.p, nrows, P64, d_z.p, kZS));
        } else {
            HIP_OK(hipMemsetAsync(d_z.p, 0, (size_t)kZS * P64 * sizeof(double), c->stream));
        }
    };
    const int zrows = hist_tiled ? 1 : kZS;
    int r = 0;
    bool chol = false;
    auto pass_update = [&](int n, int mode) {  // the p-sized half: all-reduce, then the factored update (fused.hip)
        all_reduce(c, d_z.p, (size_t)zrows * P64);
        PROFILED(c, NLE_K_SMALL,
                 nlek::sink_update(c->stream, mode, p, r, chol, d_X1.p, d_X2.p, d_lam.p, d_z.p, zrows, P64,
                                   n > 0 ? d_sAh.p + (size_t)(n - 1) * p : nullptr, NLE_EPS, d_uv.p, d_uv.p + 2 * p,
                                   d_sAh.p + (size_t)n * p, d_w.p));
    };
    pass_pixels(nlek::ROWPASS_COLSUM, nullptr);
    // factor Ka on the host while the column-sum pass runs, then upload the factors of the update
    const Nystrom ny = solve();
    r = ny.r;
    chol = ny.chol;
    f->r = r;
    f->chol_ka = ny.chol ? 1 : 0;
    f->formulation = hist ? NLE_MODE_PHI_FREE : NLE_MODE_PHI_FREE_EXP;
    build_update_operands(c, ny, p, d_X1, d_X2, d_lam);
    pass_update(0, nlek::ROWPASS_COLSUM);
    for (int n = 1; n < 2 * T; ++n) {
        pass_pixels(nlek::ROWPASS_RECIP, n == 2 * T - 1 ? d_cbuf.p : nullptr);
        pass_update(n, nlek::ROWPASS_RECIP);
    }
    // sample row sums V_A u of the scaling that defines the final c (input of the last pass) and of the
    // output of the last pass (the r scaling).  (Fetching them on a second stream, so that the Gram kernels
    // could be queued first, saved ~50 us but made two processes sharing one GPU stall for tens of
    // milliseconds per all-reduce: one stream per ctx it stays.)
    std::vector<double> sA_c(p), sA_r(p);
    HIP_OK(hipMemcpyAsync(sA_c.data(), d_sAh.p + (size_t)(2 * T - 2) * p, p * sizeof(double), hipMemcpyDeviceToHost,
                          c->stream));
    HIP_OK(hipMemcpyAsync(sA_r.data(), d_sAh.p + (size_t)(2 * T - 1) * p, p * sizeof(double), hipMemcpyDeviceToHost,
                          c->stream));
    tm_s.stop();
    tr.mark("ss: passes enqueued");
    HIP_OK(hipStreamSynchronize(c->stream));
    tr.mark("ss: sinkhorn sync");

    // Gram in sample space, enqueued; the host half that does not need it runs meanwhile.
    // Quantised luminance: histogram + fp64 GEMM over the look-up tables (k_ghist_*); otherwise
    // regenerated affinity rows on the fp64 MFMA (k_gram64).
    tm_g.start();
    const bool ghist = hist && ss.gs.nSelCols <= nlek::ghist_max_cols();
    const int ntiles = nlek::gram64_num_tiles(p);
    const size_t g_elems = ghist ? (size_t)p * p : (size_t)ntiles * 256;
    DevBuf<double> d_gpart, d_tiles(g_elems);
    auto enqueue_gram = [&] {
    if (M <= 0) {
        HIP_OK(hipMemsetAsync(d_tiles.p, 0, g_elems * sizeof(double), c->stream));
    } else if (ghist) {
        d_gpart.alloc(nlek::ghist_workspace_elems(ss.gs, nrows_local));
        static const int gmap[4] = {NLE_K_GRAM_ROWS, NLE_K_SMALL, NLE_K_GRAM_GEMM, NLE_K_SMALL};
        ProfObserver obs(c, gmap);
        HIP_OK(nlek::gram_hist(c->stream, d_lum, ss.gs, p, row0, nrows_local, d_er.p, d_ecT.p, d_Ep.p, d_cbuf.p,
                               d_gpart.p, d_tiles.p, &obs, srp));
    } else {
        d_gpart.alloc(std::max<size_t>(nlek::gram64_partial_elems(M, p), 1));
        PROFILED(c, NLE_K_GRAM, nlek::gram64(c->stream, d_lum, ss.gs, d_samples.p, p, nsw, npw, pix0, M, d_cbuf.p,
                                             d_gpart.p, d_tiles.p));
    }
    };
    OrthoSS o;
    // (the opt-in Lanczos solver works on the LITERAL q x q matrix Q = Wa + S (Wab Wab^T) S with Wa as computed, not mirrored
    // from its lower triangle -- what Spectra's DenseGenMatProd multiplies by in a USE_SPECTRA build, src/filter.cpp:174, 311
    // -- so it takes the host route, which forms exactly that; the device route diagonalises a symmetric similar matrix)
    if (ghist && std::getenv("NLE_HOST_ORTHO") == nullptr && c->topk_solver == 0) {
        // the Gram kernels were enqueued above; the q-sized products run on the device, the eigensolves on the host
        ortho_ss_device(c, o, ny, p, sA_c, sA_r, d_tiles.p, n_eig, enqueue_gram, [&] { all_reduce(c, d_tiles.p, g_elems); },
                        &ms->host, &ms->host_overlapped, tr);
        tm_g.stop();
    } else {
    enqueue_gram();
    double h0 = now_ms();
    ortho_ss_prepare(o, ny, p, sA_c, sA_r, /*literal_q=*/c->topk_solver != 0);  // host, while the Gram kernel runs
    const double h_overlapped = now_ms() - h0;
    tr.mark("ss: ortho prepare (host)");
    // (a device-to-host copy into pageable memory blocks the host until the stream reaches it, so it
    // is issued only now)
    all_reduce(c, d_tiles.p, g_elems);
    std::vector<double> tiles(g_elems);
    HIP_OK(hipMemcpyAsync(tiles.data(), d_tiles.p, tiles.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    tm_g.stop();
    HIP_OK(hipStreamSynchronize(c->stream));
    tr.mark("ss: gram sync");
    h0 = now_ms();
    ortho_ss_finish(o, ghist ? std::move(tiles) : unpack_tiles(tiles, nlek::gram64_ld(p), p, 16), n_eig, c->topk_solver);
    ms->host += now_ms() - h0;
    tr.mark("ss: ortho finish (host)");
    ms->host_overlapped += h_overlapped;
    }
    f->K = o.K;
    f->ldv = ld4(o.K);
    f->eigvals = o.Sq;
    f->r_wa = o.r_wa;
    f->r_q = o.r_q;
    f->chol_wa = o.chol_wa ? 1 : 0;

    // V = diag(c) K_AB^T D: the Nystrom extension of the K' retained eigenvectors, affinity fused
    tm_p.start();
    if (o.K > 128) throw Fail{NLE_ERR_INVALID, "Phi-free path supports at most 128 eigenvectors"};
    const int ldd = nlek::project64_ld(o.K);
    std::vector<double> Dp((size_t)p * ldd, 0.0), Vr((size_t)p * ldd, 0.0);
    for (int k = 0; k < o.K; ++k)
        for (int a = 0; a < p; ++a) {
            Dp[(size_t)a * ldd + k] = o.D[(size_t)k * p + a];
            Vr[(size_t)a * ldd + k] = o.Vrows[(size_t)k * p + a];
        }
    DevBuf<double> d_D(Dp.size());
    HIP_OK(hipMemcpyAsync(d_D.p, Dp.data(), Dp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const bool lazy = hist_tiled && std::getenv("NLE_EAGER_V") == nullptr;
    if (lazy) {
        // keep what defines V implicitly; the projection runs only if somebody asks for the matrix
        auto own = [&](auto& buf) {
            using T = std::remove_pointer_t<decltype(buf.p)>;
            const size_t bytes = buf.n * sizeof(T);
            T* ptr = buf.take();
            f->owned.emplace_back(ptr, bytes);
            return ptr;
        };
        DevBuf<double> d_Vr(Vr.size());
        DevBuf<long long> d_spix(p), d_sloc(p);
        DevBuf<float> d_slab((size_t)M);
        std::vector<long long> sloc(p);
        for (int a = 0; a < p; ++a) {
            const long long loc = ss.pix[a] - pix0;
            sloc[a] = (loc >= 0 && loc < M) ? loc : -1;
        }
        HIP_OK(hipMemcpyAsync(d_Vr.p, Vr.data(), Vr.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_spix.p, ss.pix.data(), p * sizeof(long long), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_sloc.p, sloc.data(), p * sizeof(long long), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(d_slab.p, d_lum + pix0, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));  // the host staging vectors go out of scope
        f->lazy = true;
        f->gs = ss.gs;
        f->nsw = nsw;
        f->npw = npw;
        f->ldd = ldd;
        f->P64 = P64;
        f->d_lum = own(d_slab);
        f->d_c = own(d_cbuf);
        f->d_er = own(d_er);
        f->d_ecT = own(d_ecT);
        f->d_Ep = own(d_Ep);
        f->d_D = own(d_D);
        f->d_Vrows = own(d_Vr);
        f->d_samples = own(d_samples);
        f->d_sample_pix = own(d_spix);
        f->d_sample_loc = own(d_sloc);
        if (sorted) {
            f->has_sorted = true;
            f->sorted = nlek::SortedRows{own(d_scol), own(d_desc), own(d_first), own(d_E), sr.rec, sr.kappa};
            f->sorted.lev_t0 = sr.lev_t0;
            f->sorted.lev_nt = sr.lev_nt;
            f->sorted.mom = sr.mom;
        }
        f->h_Vrows = o.Vrows;
        f->h_sample_pix = ss.pix;
        tm_p.stop();
        ms->sinkhorn = tm_s.ms();
        ms->gram = tm_g.ms();
        ms->project = tm_p.ms();
        return;
    }
    DevBuf<float> d_V((size_t)std::max<long long>(M, 1) * f->ldv);
    if (hist && nlek::project_hist_ok(ss.gs, p, o.K) && std::getenv("NLE_PROJECT_HIST") != nullptr)
        PROFILED(c, NLE_K_PROJECT, nlek::project_hist(c->stream, d_lum, ss.gs, p, row0, nrows_local, d_er.p, d_ecT.p,
                                                      d_Ep.p, d_D.p, ldd, o.K, d_cbuf.p, d_V.p, f->ldv));
    else
        PROFILED(c, NLE_K_PROJECT, nlek::project64(c->stream, d_lum, ss.gs, d_samples.p, p, nsw, npw, pix0, M, d_D.p, o.K,
                                                   d_cbuf.p, d_V.p, f->ldv));
    tr.mark("ss: project enqueued");
    scatter_sample_rows(c, ss.pix, p, o.Vrows, p, o.K, f->ldv, pix0, M, d_V.p);
    tm_p.stop();
    HIP_OK(hipStreamSynchronize(c->stream));
    tr.mark("ss: project sync");
    f->v_bytes = d_V.n * sizeof(float);
    f->d_V = d_V.take();
    ms->sinkhorn = tm_s.ms();
    ms->gram = tm_g.ms();
    ms->project = tm_p.ms();
}

// (3) The same sample-space algebra on fp64 affinity rows (k_affinity64: libm exp of the reference's own argument, :104-112),
// regenerated CHUNK BY CHUNK in every pass: no N x r matrix, a bounded workspace, any luminance plane, any grid up to 2048
// samples, any K.  Every N-sized step is a generic64.hip kernel on the chunk (k_i = row i of the chunk):
//   Sinkhorn half-iteration   y_i = recip(k_i . w), z += k_i y_i                 k_rowpass64 (u := w)
//   Gram                      Gk += sum c_i^2 k_i k_i^T                          k_gram64d
//   eigenvectors              V_i = c_i k_i^T D                                  k_tsgemm64      (V: N x K' fp64, as mode 4)
// and the p-sized side is train_sample_space's (factored update, ortho_ss_device).  Costs a pass 2 x N p 8 bytes of HBM
// traffic (write + read of the chunk) where the materialised form reads N r 8 once -- the price of not holding it.
void train_stream64(nle_ctx* c, nle_filter* f, const float* d_lum, const SampleSet& ss, const std::function<Nystrom()>& solve,
                    double hx, double hy, int T, int n_eig, long long pix0, long long M, StageMs* ms) {
    const int p = ss.p, ld = ld4(p);
    const double sw = 1.0 / (hx * hx), pw = 1.0 / (hy * hy);
    hipStream_t st = c->stream;
    Trace tr;
    Timer tm_s(st), tm_g(st), tm_p(st);
    tm_s.start();
    size_t budget_mb = 2048;
    if (const char* e = std::getenv("NLE_STREAM64_CHUNK_MB")) budget_mb = (size_t)std::max(1, std::atoi(e));
    const long long rows_fit = (long long)((budget_mb << 20) / ((size_t)ld * sizeof(double)));
    const long long CH = std::max<long long>(256, std::min<long long>(std::max<long long>(M, 1), rows_fit));
    DevBuf<float4> d_samples(p);
    HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), p * sizeof(float4), hipMemcpyHostToDevice, st));
    DevBuf<double> d_K((size_t)CH * ld), d_partial((size_t)nlek::kRowpassMaxBlocks * ld), d_zc(ld), d_z(ld), d_w(ld), d_ones(ld),
        d_sAh((size_t)2 * T * p), d_X1, d_X2, d_lam, d_uv((size_t)3 * p), d_cbuf((size_t)std::max<long long>(M, 1));
    HIP_OK(hipMemsetAsync(d_w.p, 0, ld * sizeof(double), st));
    HIP_OK(nlek::fill64(st, d_ones.p, ld, 1.0));
    tr.mark("s64: alloc+upload");
    auto chunk_rows = [&](long long i0) { return std::min<long long>(CH, M - i0); };
    auto gen = [&](long long i0, long long mc) {
        PROFILED(c, NLE_K_AFFINITY, nlek::affinity64(st, d_lum, ss.gs, d_samples.p, p, ld, sw, pw, pix0 + i0, mc, d_K.p, true));
    };
    auto pass_pixels = [&](int mode, double* cbuf) {
        HIP_OK(hipMemsetAsync(d_z.p, 0, ld * sizeof(double), st));
        for (long long i0 = 0; i0 < M; i0 += CH) {
            const long long mc = chunk_rows(i0);
            gen(i0, mc);
            int nb = 0;
            PROFILED(c, NLE_K_SINKHORN_PASS, nlek::rowpass64(st, mode, d_K.p, mc, ld, d_w.p, d_ones.p, nullptr, NLE_EPS, d_partial.p, &nb));
            PROFILED(c, NLE_K_REDUCE, nlek::reduce_partials(st, d_partial.p, nb, ld, d_zc.p));
            HIP_OK(nlek::add64(st, d_z.p, d_zc.p, ld));
            if (cbuf) PROFILED(c, NLE_K_SMALL, nlek::row_scalings64(st, d_K.p, mc, ld, p, d_w.p, NLE_EPS, cbuf + i0));
        }
    };
    int r = 0;
    bool chol = false;
    auto pass_update = [&](int n, int mode) {
        all_reduce(c, d_z.p, (size_t)ld);
        PROFILED(c, NLE_K_SMALL, nlek::sink_update(st, mode, p, r, chol, d_X1.p, d_X2.p, d_lam.p, d_z.p, 1, ld,
                                                   n > 0 ? d_sAh.p + (size_t)(n - 1) * p : nullptr, NLE_EPS, d_uv.p, d_uv.p + 2 * p,
                                                   d_sAh.p + (size_t)n * p, d_w.p));
    };
    pass_pixels(nlek::ROWPASS_COLSUM, nullptr);
    const Nystrom ny = solve();
    r = ny.r;
    chol = ny.chol;
    f->r = r;
    f->chol_ka = ny.chol ? 1 : 0;
    f->formulation = NLE_MODE_STREAMED_F64;
    build_update_operands(c, ny, p, d_X1, d_X2, d_lam);
    pass_update(0, nlek::ROWPASS_COLSUM);
    for (int n = 1; n < 2 * T; ++n) {
        pass_pixels(nlek::ROWPASS_RECIP, n == 2 * T - 1 ? d_cbuf.p : nullptr);
        pass_update(n, nlek::ROWPASS_RECIP);
    }
    std::vector<double> sA_c(p), sA_r(p);
    HIP_OK(hipMemcpyAsync(sA_c.data(), d_sAh.p + (size_t)(2 * T - 2) * p, p * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(sA_r.data(), d_sAh.p + (size_t)(2 * T - 1) * p, p * sizeof(double), hipMemcpyDeviceToHost, st));
    tm_s.stop();
    HIP_OK(hipStreamSynchronize(st));
    tr.mark("s64: sinkhorn");
    // Gram: Gk = sum over the non-sample pixels of c_i^2 k_i k_i^T, chunk by chunk
    tm_g.start();
    const size_t pp = (size_t)p * p;
    DevBuf<double> d_G(pp), d_Gc(pp), d_gpart(std::max<size_t>(nlek::gram64d_partial_elems(CH, p), 1));
    auto enqueue_gram = [&] {
        HIP_OK(hipMemsetAsync(d_G.p, 0, pp * sizeof(double), st));
        for (long long i0 = 0; i0 < M; i0 += CH) {
            const long long mc = chunk_rows(i0);
            gen(i0, mc);
            PROFILED(c, NLE_K_GRAM, nlek::gram64d(st, d_K.p, mc, ld, p, d_cbuf.p + i0, d_gpart.p, d_Gc.p));
            HIP_OK(nlek::add64(st, d_G.p, d_Gc.p, pp));
        }
    };
    OrthoSS o;
    ortho_ss_device(c, o, ny, p, sA_c, sA_r, d_G.p, n_eig, enqueue_gram, [&] { all_reduce(c, d_G.p, pp); }, &ms->host,
                    &ms->host_overlapped, tr);
    tm_g.stop();
    f->K = o.K;
    f->ldv = ld4(o.K);
    f->eigvals = o.Sq;
    f->r_wa = o.r_wa;
    f->r_q = o.r_q;
    f->chol_wa = o.chol_wa ? 1 : 0;
    // V = diag(c) K D (the Nystrom extension of the K' kept eigenvectors, :324-327) + the exact sample rows
    tm_p.start();
    DevBuf<double> d_D((size_t)p * o.K), d_V((size_t)std::max<long long>(M, 1) * f->ldv);
    HIP_OK(hipMemcpyAsync(d_D.p, o.D.data(), o.D.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(d_V.p, 0, d_V.n * sizeof(double), st));
    for (long long i0 = 0; i0 < M; i0 += CH) {
        const long long mc = chunk_rows(i0);
        gen(i0, mc);
        PROFILED(c, NLE_K_PROJECT, nlek::ts_gemm64(st, d_K.p, mc, ld, p, d_D.p, o.K, d_cbuf.p + i0, d_V.p + (size_t)i0 * f->ldv, f->ldv));
    }
    {
        std::vector<double> rows;
        std::vector<long long> idx;
        for (int a = 0; a < p; ++a) {
            const long long loc = ss.pix[a] - pix0;
            if (loc < 0 || loc >= M) continue;
            idx.push_back(loc);
            const size_t off = rows.size();
            rows.resize(off + f->ldv, 0.0);
            for (int k = 0; k < o.K; ++k) rows[off + k] = o.Vrows[(size_t)k * p + a];
        }
        if (!idx.empty()) {
            DevBuf<double> d_rows(rows.size());
            DevBuf<long long> d_idx(idx.size());
            HIP_OK(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_OK(hipMemcpyAsync(d_idx.p, idx.data(), idx.size() * sizeof(long long), hipMemcpyHostToDevice, st));
            PROFILED(c, NLE_K_SMALL, nlek::scatter_rows64(st, d_rows.p, d_idx.p, (int)idx.size(), f->ldv, d_V.p, M));
            HIP_OK(hipStreamSynchronize(st));
        }
    }
    tm_p.stop();
    HIP_OK(hipStreamSynchronize(st));
    tr.mark("s64: project");
    f->v64_bytes = d_V.n * sizeof(double);
    f->d_V64 = d_V.take();
    ms->sinkhorn = tm_s.ms();
    ms->gram = tm_g.ms();
    ms->project = tm_p.ms();
}

// materialise V = diag(c) K D of a lazy filter (projection kernel + exact sample rows)
void ensure_V(nle_filter* f) {
    if (f->d_V) return;
    nle_ctx* c = f->ctx;
    if (f->d_V64) {  // fp64 formulation: an fp32 copy for the accessors that hand out float pointers
        DevBuf<float> d_V((size_t)std::max<long long>(f->n_local, 1) * f->ldv);
        HIP_OK(nlek::to_f32(c->stream, f->d_V64, f->n_local * f->ldv, d_V.p));
        HIP_OK(hipStreamSynchronize(c->stream));
        f->v_bytes = d_V.n * sizeof(float);
        f->d_V = d_V.take();
        return;
    }
    if (!f->lazy) return;
    const long long M = f->n_local, pix0 = (long long)f->row0 * f->W;
    DevBuf<float> d_V((size_t)std::max<long long>(M, 1) * f->ldv);
    PROFILED(c, NLE_K_PROJECT, nlek::project64(c->stream, f->d_lum - pix0, f->gs, f->d_samples, f->p, f->nsw, f->npw, pix0,
                                               M, f->d_D, f->K, f->d_c, d_V.p, f->ldv));
    scatter_sample_rows(c, f->h_sample_pix, f->p, f->h_Vrows, f->p, f->K, f->ldv, pix0, M, d_V.p);
    HIP_OK(hipStreamSynchronize(c->stream));
    f->v_bytes = d_V.n * sizeof(float);
    f->d_V = d_V.take();
}

// apply on the p-sized side of a lazy filter: reduce half (column sums m = sum_i k_i c_i x_i through the
// tables), the p/K-sized middle (k_apply_small), and one table pass per output layer
// `done(l0, nl)`, when given, is called after layers [l0, l0 + nl) are complete on the stream (the host-buffer entry
// points start their download there); `group` caps the layers per launch (0: as many as fit)
using LayersDone = std::function<void(int, int)>;
void apply_sample_space(nle_filter* f, const float* d_x, const double* h_g /* L x K */, int L, float* d_y,
                        const LayersDone& done = nullptr, int group = 0, bool round8 = false) {
    nle_ctx* c = f->ctx;
    const long long M = f->n_local, pix0 = (long long)f->row0 * f->W;
    const int p = f->p, K = f->K, P64 = f->P64, nrows_local = (int)(M / f->W);
    const float* lum = f->d_lum - pix0;  // indexed by global pixel, only this rank's rows are touched
    DevBuf<double> d_ws(std::max<size_t>(nlek::hist_tiled_workspace_elems(f->gs, std::max(nrows_local, 1)), 1)), d_m(P64),
        d_resp((size_t)L * K), d_t(K), d_Wp((size_t)L * P64), d_YA((size_t)L * p);
    HIP_OK(hipMemcpyAsync(d_resp.p, h_g, (size_t)L * K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (M > 0) {
        static const int rmap[4] = {NLE_K_SINK_TABLES, NLE_K_APPLY_REDUCE, NLE_K_REDUCE, NLE_K_REDUCE};
        ProfObserver obs(c, rmap);
        HIP_OK(nlek::sink_hist_tiled(c->stream, nlek::ROWPASS_XVEC, lum, f->gs, p, P64, f->row0, nrows_local, f->d_er,
                                     f->d_ecT, f->d_Ep, nullptr, NLE_EPS, nullptr, d_ws.p, d_m.p, &obs, f->d_c, d_x,
                                     f->has_sorted ? &f->sorted : nullptr));
    } else {
        HIP_OK(hipMemsetAsync(d_m.p, 0, P64 * sizeof(double), c->stream));
    }
    all_reduce(c, d_m.p, P64);
    // x at the p sample pixels: every rank's own rows, completed by the all-reduce when the planes are slabs
    DevBuf<double> d_xA(p);
    {
        const bool slabs = c->slab_input && c->world > 1;
        PROFILED(c, NLE_K_SMALL, nlek::gather_samples_slab(c->stream, d_x, f->gs, slabs ? f->row0 : 0, slabs ? f->row1 : f->H, d_xA.p));
        if (slabs) all_reduce(c, d_xA.p, p);
    }
    PROFILED(c, NLE_K_SMALL, nlek::apply_small(c->stream, p, K, f->ldd, L, P64, d_m.p, f->d_D, f->d_Vrows, d_xA.p,
                                               d_resp.p, d_t.p, d_Wp.p, d_YA.p));
    if (M > 0) {
        static const int emap[4] = {NLE_K_SINK_TABLES, NLE_K_APPLY_EXPAND, NLE_K_REDUCE, NLE_K_REDUCE};
        int lb = std::min(L, nlek::apply_layers_per_launch(f->gs));
        if (f->has_sorted && f->gs.nSelCols <= nlek::sorted_expand_max_cols() && f->gs.W <= nlek::sorted_expand_max_width() &&
            std::getenv("NLE_NO_SORTED_EXPAND") == nullptr)
            lb = std::min(L, nlek::sorted_expand_layers(f->gs));  // what one launch of the sorted expand kernel takes
        if (group > 0) lb = std::min(lb, group);
        DevBuf<double> d_gws((size_t)lb * nrows_local * 256 * f->gs.nSelCols);
        for (int l = 0; l < L; l += lb) {
            const int nl = std::min(lb, L - l);
            {
                ProfObserver obs(c, emap);
                HIP_OK(nlek::apply_hist_layers(c->stream, lum, f->gs, p, f->row0, nrows_local, f->d_er, f->d_ecT, f->d_Ep,
                                               d_Wp.p + (size_t)l * P64, P64, nl, f->d_c, d_gws.p, d_y + (size_t)l * M, M,
                                               &obs, f->has_sorted ? &f->sorted : nullptr, round8));
            }
            PROFILED(c, NLE_K_SMALL, nlek::scatter_samples(c->stream, p, nl, f->d_sample_loc, d_YA.p + (size_t)l * p,
                                                           d_y + (size_t)l * M, M, round8));
            if (done) done(l, nl);
        }
    } else if (done) {
        done(0, L);
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    prof_flush(c);
}

// d_lum_in: the full H x W plane, or -- ctx in slab-input mode -- this rank's rows [row0, row1) only
nle_filter* train_impl(nle_ctx* c, const float* d_lum_in, int H, int W, int nRow, int nCol, double hx,
                       double hy, int T, int n_eig) {
    check_image_size(H, W);
    const float* d_lum = d_lum_in;
    if (c->slab_input && c->world > 1) {  // virtual base of the full image: only this rank's rows are ever dereferenced
        int r0, r1;
        slab(H, c->rank, c->world, &r0, &r1);
        d_lum = d_lum_in - (size_t)r0 * W;
    }
    if (nRow > H || nCol > W)  // reference src/filter.cpp:117-119
        throw Fail{NLE_ERR_INVALID, "Number of samples per row and col must be <= that of image."};
    GridSpec gs;
    if (!make_grid(H, W, nRow, nCol, &gs)) throw Fail{NLE_ERR_INVALID, "invalid sample counts"};
    if (T < 1) throw Fail{NLE_ERR_INVALID, "nSinkhornIter must be >= 1"};
    if (n_eig < 1) throw Fail{NLE_ERR_INVALID, "nEigenVectors must be >= 1"};
    if (!(hx > 0) || !(hy > 0)) throw Fail{NLE_ERR_INVALID, "hx and hy must be > 0"};
    if (gs.p() > 2048) throw Fail{NLE_ERR_INVALID, "more than 2048 samples is not supported"};
    // every rank owns at least one image row: the formulation, the collective sizes and their order are then the
    // same on all ranks (an empty slab used to take a different path and mismatch the all-reduces)
    if (c->world > H) throw Fail{NLE_ERR_INVALID, "more ranks than image rows"};
    // Phi-free needs <= 128 eigenvectors; its generic kernels need <= 256 samples, its table kernels
    // (quantised luminance, checked on the device below) a sample grid of at most 32 x 36.
    const bool generic_ok = gs.p() <= nlek::sink_pass_max_p() && n_eig <= 128;
    const bool tables_ok = n_eig <= 128 && gs.nSelCols <= nlek::ghist_max_cols() && gs.nSelRows <= 32 && c->mode != 3;
    if (c->mode == 3 && !generic_ok)
        throw Fail{NLE_ERR_INVALID, "Phi-free path without tables supports at most 256 samples and 128 eigenvectors"};
    if (c->mode == 2 && !generic_ok && !tables_ok)
        throw Fail{NLE_ERR_INVALID, "Phi-free path supports at most 128 eigenvectors and a 32 x 36 sample grid"};
    // auto: the table form (all fp64) whenever it applies, else the literal decomposition in fp64 (generic64.hip).  The
    // fp32 formulations (materialised Phi, Phi-free with fp32 affinities) run only when asked for by mode: they miss
    // the 1e-4 bar on some well-posed inputs (DESIGN.md "Numerics").
    const bool want_fuse = c->mode == 2 || c->mode == 3 || (c->mode == 0 && tables_ok);
    HIP_OK(hipSetDevice(c->device));

    auto f = new nle_filter();
    try {
        f->ctx = c;
        f->H = H;
        f->W = W;
        slab(H, c->rank, c->world, &f->row0, &f->row1);
        const long long pix0 = (long long)f->row0 * W;
        const long long M = (long long)(f->row1 - f->row0) * W;
        f->n_local = M;
        const double t_begin = now_ms();
        pinned_reset(c);
        Trace tr;
        StageMs sm;
        // --- sample set, Ka and its eigenpairs (:486-491, host fp64)
        Timer tm_a(c->stream);
        tm_a.start();
        SampleSet ss = fetch_samples(c, d_lum, gs, want_fuse && tables_ok, c->slab_input && c->world > 1);
        const bool fuse = c->mode == 0 ? (tables_ok && ss.quantised)
                                       : (want_fuse && (generic_ok || (tables_ok && ss.quantised)));
        if (c->mode == 2 && !fuse)
            throw Fail{NLE_ERR_INVALID, "Phi-free path: more than 256 samples needs an integer-valued luminance plane"};
        tr.mark("fetch_samples");
        f->p = ss.p;
        double h0 = now_ms();
        std::vector<double> Ka = build_Ka(ss, hx, hy);
        tr.mark("build_Ka");
        sm.host += now_ms() - h0;
        auto solve = [&](bool allow_chol) {
            const double t0 = now_ms();
            Nystrom ny = solve_Ka(c, Ka, ss.p, allow_chol);
            tr.mark(ny.chol ? "chol(Ka)" : "eig(Ka)");
            sm.host += now_ms() - t0;
            return ny;
        };
        tm_a.stop();
        // auto mode's fp64 fallback holds Phi (N x r doubles) when that fits comfortably (a pass reads it once); otherwise
        // -- and when asked for -- the streamed form, which holds nothing N x r (the ranks agree on it: ranks_where)
        bool stream64 = c->mode == NLE_MODE_STREAMED_F64;
        if (!fuse && c->mode == NLE_MODE_AUTO) {
            size_t free_b = 0, total_b = 0;
            const size_t need = (size_t)std::max<long long>(M, 1) * ld4(ss.p) * sizeof(double);
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (free_b + c->arena_bytes) / 2) stream64 = true;
            if (std::getenv("NLE_AUTO_STREAM64")) stream64 = true;
            if (c->world > 1) stream64 = ranks_where(c, stream64) > 0;  // one rank short of memory: everybody streams
        }
        if (fuse) {
            train_sample_space(c, f, d_lum, ss, [&] { return solve(true); }, hx, hy, T, n_eig, pix0, M, &sm);
        } else if (stream64) {
            train_stream64(c, f, d_lum, ss, [&] { return solve(true); }, hx, hy, T, n_eig, pix0, M, &sm);
        } else {
            const Nystrom ny = solve(false);
            f->r = ny.r;
            if (c->mode == NLE_MODE_MATERIALISED)
                train_materialised(c, f, d_lum, ss, ny, hx, hy, T, n_eig, pix0, M, &sm);
            else
                train_generic64(c, f, d_lum, ss, ny, hx, hy, T, n_eig, pix0, M, &sm);
        }
        tr.mark("train path");
        prof_flush(c);
        f->ms[0] = tm_a.ms();
        f->ms[1] = sm.sinkhorn;
        f->ms[2] = sm.gram;
        f->ms[3] = sm.project;
        f->ms[4] = sm.host;
        f->ms[5] = now_ms() - t_begin;
        c->filters.insert(f);
    } catch (...) {
        delete f;
        throw;
    }
    return f;
}

// t = V^T x (all ranks), then Y[l] = V (g_l o t)
// round8: the planes come out clamped to [0, 255] and rounded half to even (src/filter.cpp:434-436) -- on the default path from
// the fp64 value, before anything is rounded to fp32 (other formulations: their fp32 planes, rounded by the caller)
void apply_impl(nle_filter* f, const float* d_x_in, int H, int W, const double* h_g /* L x K */, int L,
                float* d_y, const LayersDone& done = nullptr, int group = 0, bool round8 = false) {
    nle_ctx* c = f->ctx;
    // slab-input mode: d_x_in holds this rank's rows only; index it through the virtual base of the full image
    const float* d_x = (c->slab_input && c->world > 1) ? d_x_in - (size_t)f->row0 * f->W : d_x_in;
    if ((long long)H * W != (long long)f->H * f->W)  // reference src/filter.cpp:447-449
        throw Fail{NLE_ERR_INVALID, "Number of values in channel must match that of training image."};
    if (L < 1 || L > 64) throw Fail{NLE_ERR_INVALID, "number of layers must be in [1, 64]"};
    HIP_OK(hipSetDevice(c->device));
    if (f->lazy && std::getenv("NLE_APPLY_WITH_V") == nullptr) {
        apply_sample_space(f, d_x, h_g, L, d_y, done, group, round8);
        return;
    }
    if (!f->d_V64) ensure_V(f);
    const int ld = f->ldv;
    const long long M = f->n_local;
    const long long pix0 = (long long)f->row0 * f->W;
    DevBuf<double> d_partial((size_t)nlek::kRowpassMaxBlocks * ld), d_t(ld), d_resp((size_t)L * ld),
        d_g((size_t)L * ld);
    std::vector<double> resp((size_t)L * ld, 0.0);
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < f->K; ++k) resp[(size_t)l * ld + k] = h_g[(size_t)l * f->K + k];
    HIP_OK(hipMemcpyAsync(d_resp.p, resp.data(), resp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int nb = 0;
    if (f->d_V64)
        PROFILED(c, NLE_K_APPLY_REDUCE, nlek::rowpass64(c->stream, nlek::ROWPASS_XVEC, f->d_V64, M, ld, nullptr, nullptr,
                                                        d_x + pix0, NLE_EPS, d_partial.p, &nb));
    else
    PROFILED(c, NLE_K_APPLY_REDUCE, nlek::rowpass(c->stream, nlek::ROWPASS_XVEC, f->d_V, M, ld, nullptr, nullptr,
                                                  d_x + pix0, NLE_EPS, d_partial.p, &nb));
    PROFILED(c, NLE_K_REDUCE, nlek::reduce_partials(c->stream, d_partial.p, nb, ld, d_t.p));
    all_reduce(c, d_t.p, ld);
    for (int l = 0; l < L; ++l)
        PROFILED(c, NLE_K_SMALL, nlek::scale_vec(c->stream, d_resp.p + (size_t)l * ld, d_t.p, ld, d_g.p + (size_t)l * ld));
    if (f->d_V64)
        PROFILED(c, NLE_K_APPLY_EXPAND, nlek::apply_expand64(c->stream, f->d_V64, M, ld, f->K, d_g.p, L, d_y, M));
    else
    PROFILED(c, NLE_K_APPLY_EXPAND, nlek::apply_expand(c->stream, f->d_V, M, ld, d_g.p, L, d_y, M));
    if (done) done(0, L);
    HIP_OK(hipStreamSynchronize(c->stream));
    prof_flush(c);
}

void layer_resp(const double* ev, int K, int L, double* out) {
    // detail layer j <-> lambda^j - lambda^(j+1); base <-> lambda^(L-1)  (reference :334-347)
    for (int j = 0; j < L; ++j)
        for (int k = 0; k < K; ++k) {
            const double a = std::pow(ev[k], (double)j);
            out[(size_t)j * K + k] = (j < L - 1) ? a - std::pow(ev[k], (double)(j + 1)) : a;
        }
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI
extern "C" {

int nle_layer_responses(const double* h_eigvals, int K, int L, double* h_resp) {
    if (!h_eigvals || !h_resp || K < 0 || L < 1) return NLE_ERR_INVALID;
    layer_resp(h_eigvals, K, L, h_resp);
    return NLE_OK;
}

int nle_compute_kernel(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples,
                       double hx, double hy, double* h_Ka, float* d_kab) {
    if (!ctx || !d_lum) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        check_image_size(H, W);
        if (n_row_samples > H || n_col_samples > W)
            throw Fail{NLE_ERR_INVALID, "Number of samples per row and col must be <= that of image."};
        GridSpec gs;
        if (!make_grid(H, W, n_row_samples, n_col_samples, &gs)) throw Fail{NLE_ERR_INVALID, "invalid sample counts"};
        HIP_OK(hipSetDevice(ctx->device));
        SampleSet ss = fetch_samples(ctx, d_lum, gs);
        if (h_Ka) {
            std::vector<double> Ka = build_Ka(ss, hx, hy);
            std::copy(Ka.begin(), Ka.end(), h_Ka);
        }
        if (d_kab) {
            int row0, row1;
            slab(H, ctx->rank, ctx->world, &row0, &row1);
            DevBuf<float4> d_samples(ss.p);
            HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), ss.p * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
            PROFILED(ctx, NLE_K_AFFINITY,
                     nlek::affinity(ctx->stream, d_lum, gs, d_samples.p, ss.p, ld4(ss.p), nsw_of(hx), nsw_of(hy),
                                    (long long)row0 * W, (long long)(row1 - row0) * W, d_kab));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            prof_flush(ctx);
        }
    });
}

int nle_nystrom(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                double hy, double* h_eigvals, int* r, float* d_phi) {
    if (!ctx || !d_lum || !d_phi || !r) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        check_image_size(H, W);
        if (n_row_samples > H || n_col_samples > W)
            throw Fail{NLE_ERR_INVALID, "Number of samples per row and col must be <= that of image."};
        GridSpec gs;
        if (!make_grid(H, W, n_row_samples, n_col_samples, &gs)) throw Fail{NLE_ERR_INVALID, "invalid sample counts"};
        HIP_OK(hipSetDevice(ctx->device));
        SampleSet ss = fetch_samples(ctx, d_lum, gs);
        std::vector<double> Ka = build_Ka(ss, hx, hy);
        Nystrom ny = solve_Ka(nullptr, Ka, ss.p, false);
        int row0, row1;
        slab(H, ctx->rank, ctx->world, &row0, &row1);
        build_phi(ctx, d_lum, ss, ny, hx, hy, (long long)row0 * W, (long long)(row1 - row0) * W, d_phi);
        *r = ny.r;
        if (h_eigvals) std::copy(ny.lam.begin(), ny.lam.end(), h_eigvals);
    });
}

int nle_ts_gemm(nle_ctx* ctx, const float* d_A, long long M, int lda, int kd, const double* h_B, int nc, float* d_C) {
    if (!ctx || !d_A || !h_B || !d_C || M < 0 || kd < 1 || nc < 1 || lda < kd || (lda & 3)) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        const int ldc = ld4(nc);
        std::vector<float> B((size_t)kd * ldc, 0.f);
        for (int j = 0; j < nc; ++j)
            for (int k = 0; k < kd; ++k) B[(size_t)k * ldc + j] = (float)h_B[(size_t)j * kd + k];
        DevBuf<float> d_B(B.size());
        HIP_OK(hipMemcpyAsync(d_B.p, B.data(), B.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        GridSpec gs{};
        HIP_OK(nlek::ts_gemm(ctx->stream, false, d_A, lda, nullptr, gs, nullptr, 0.f, 0.f, 0, d_B.p, ldc, kd, d_C, ldc,
                             M, nullptr, NLE_EPS));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_sinkhorn_scalings(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r, const double* h_eigvals,
                          int max_iter, double* h_u_c, double* h_u_r) {
    if (!ctx || !d_phi || !h_eigvals || !h_u_c || !h_u_r || M < 0 || r < 1 || ld < r || (ld & 3)) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        std::vector<double> lam(h_eigvals, h_eigvals + r), uc, ur;
        sinkhorn_passes(ctx, d_phi, M, ld, r, lam, max_iter, &uc, &ur, nullptr);
        std::copy(uc.begin(), uc.end(), h_u_c);
        std::copy(ur.begin(), ur.end(), h_u_r);
    });
}

int nle_gram(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r, const double* h_u, double* h_G) {
    if (!ctx || !d_phi || !h_G || M < 0 || r < 1 || ld < r || (ld & 3)) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        DevBuf<double> d_u;
        if (h_u) {
            std::vector<double> u(ld, 0.0);
            std::copy(h_u, h_u + r, u.begin());
            d_u.alloc(ld);
            HIP_OK(hipMemcpy(d_u.p, u.data(), ld * sizeof(double), hipMemcpyHostToDevice));
        }
        std::vector<double> G = gram_all(ctx, d_phi, M, ld, r, d_u.p);
        std::copy(G.begin(), G.end(), h_G);
    });
}

int nle_row_scalings(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r, const double* h_u, double* d_out) {
    if (!ctx || !d_phi || !h_u || !d_out || M < 0 || r < 1 || ld < r || (ld & 3)) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        std::vector<double> u(ld, 0.0);
        std::copy(h_u, h_u + r, u.begin());
        DevBuf<double> d_u(ld);
        HIP_OK(hipMemcpyAsync(d_u.p, u.data(), ld * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(nlek::row_scalings(ctx->stream, d_phi, M, ld, d_u.p, NLE_EPS, d_out));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

// ---- the same five stage entry points on fp64 device matrices (generic64.hip) ----
int nle_compute_kernel64(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                         double hy, double* h_Ka, double* d_kab) {
    if (!ctx || !d_lum) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        check_image_size(H, W);
        if (n_row_samples > H || n_col_samples > W)
            throw Fail{NLE_ERR_INVALID, "Number of samples per row and col must be <= that of image."};
        GridSpec gs;
        if (!make_grid(H, W, n_row_samples, n_col_samples, &gs)) throw Fail{NLE_ERR_INVALID, "invalid sample counts"};
        HIP_OK(hipSetDevice(ctx->device));
        SampleSet ss = fetch_samples(ctx, d_lum, gs);
        if (h_Ka) {
            std::vector<double> Ka = build_Ka(ss, hx, hy);
            std::copy(Ka.begin(), Ka.end(), h_Ka);
        }
        if (d_kab) {
            int row0, row1;
            slab(H, ctx->rank, ctx->world, &row0, &row1);
            DevBuf<float4> d_samples(ss.p);
            HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), ss.p * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
            PROFILED(ctx, NLE_K_AFFINITY,
                     nlek::affinity64(ctx->stream, d_lum, gs, d_samples.p, ss.p, ld4(ss.p), 1.0 / (hx * hx), 1.0 / (hy * hy),
                                      (long long)row0 * W, (long long)(row1 - row0) * W, d_kab));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            prof_flush(ctx);
        }
    });
}

int nle_ts_gemm64(nle_ctx* ctx, const double* d_A, long long M, int lda, int kd, const double* h_B, int nc, double* d_C) {
    if (!ctx || !d_A || !h_B || !d_C || M < 0 || kd < 1 || nc < 1 || lda < kd) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        const int ldc = ld4(nc);
        DevBuf<double> d_B((size_t)kd * nc);
        HIP_OK(hipMemcpyAsync(d_B.p, h_B, (size_t)kd * nc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemsetAsync(d_C, 0, (size_t)std::max<long long>(M, 0) * ldc * sizeof(double), ctx->stream));
        HIP_OK(nlek::ts_gemm64(ctx->stream, d_A, M, lda, kd, d_B.p, nc, nullptr, d_C, ldc));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_sinkhorn_scalings64(nle_ctx* ctx, const double* d_phi, long long M, int ld, int r, const double* h_eigvals,
                            int max_iter, double* h_u_c, double* h_u_r) {
    if (!ctx || !d_phi || !h_eigvals || !h_u_c || !h_u_r || M < 0 || r < 1 || ld < r) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        std::vector<double> lam(h_eigvals, h_eigvals + r), uc, ur;
        sinkhorn_passes(ctx, d_phi, M, ld, r, lam, max_iter, &uc, &ur, nullptr);
        std::copy(uc.begin(), uc.end(), h_u_c);
        std::copy(ur.begin(), ur.end(), h_u_r);
    });
}

int nle_gram64(nle_ctx* ctx, const double* d_phi, long long M, int ld, int r, const double* h_u, double* h_G) {
    if (!ctx || !d_phi || !h_G || M < 0 || r < 1 || ld < r) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        DevBuf<double> d_u;
        if (h_u) {
            d_u.alloc(r);
            HIP_OK(hipMemcpy(d_u.p, h_u, r * sizeof(double), hipMemcpyHostToDevice));
        }
        std::vector<double> G = gram_all64(ctx, d_phi, M, ld, r, d_u.p);
        std::copy(G.begin(), G.end(), h_G);
    });
}

int nle_row_scalings64(nle_ctx* ctx, const double* d_phi, long long M, int ld, int r, const double* h_u, double* d_out) {
    if (!ctx || !d_phi || !h_u || !d_out || M < 0 || r < 1 || ld < r) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        DevBuf<double> d_u(r);
        HIP_OK(hipMemcpyAsync(d_u.p, h_u, r * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(nlek::row_scalings64(ctx->stream, d_phi, M, ld, r, d_u.p, NLE_EPS, d_out));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_train(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
              double hy, int n_sinkhorn_iter, int n_eigen_vectors, nle_filter** out) {
    if (!ctx || !d_lum || !out) return NLE_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        *out = train_impl(ctx, d_lum, H, W, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter, n_eigen_vectors);
    });
}

int nle_train_host(nle_ctx* ctx, const float* h_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                   double hy, int n_sinkhorn_iter, int n_eigen_vectors, nle_filter** out) {
    if (!ctx || !h_lum || !out) return NLE_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        check_image_size(H, W);
        HIP_OK(hipSetDevice(ctx->device));
        size_t npx = (size_t)H * W;  // slab-input mode: h_lum holds this rank's rows only
        if (ctx->slab_input && ctx->world > 1) {
            int r0, r1;
            slab(H, ctx->rank, ctx->world, &r0, &r1);
            npx = (size_t)(r1 - r0) * W;
        }
        DevBuf<float> d_lum(std::max<size_t>(npx, 1));
        HIP_OK(hipMemcpyAsync(d_lum.p, h_lum, npx * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        nle_filter* f = train_impl(ctx, d_lum.p, H, W, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter, n_eigen_vectors);
        f->plane_bytes = d_lum.n * sizeof(float);  // kept: nle_apply*_host(h_x == NULL) filters the training plane
        f->d_plane = d_lum.take();
        *out = f;
    });
}

int nle_train_host_u8(nle_ctx* ctx, const unsigned char* h_lum8, int H, int W, int n_row_samples, int n_col_samples, double hx,
                      double hy, int n_sinkhorn_iter, int n_eigen_vectors, nle_filter** out) {
    if (!ctx || !h_lum8 || !out) return NLE_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        check_image_size(H, W);
        HIP_OK(hipSetDevice(ctx->device));
        size_t npx = (size_t)H * W;  // slab-input mode: h_lum8 holds this rank's rows only
        if (ctx->slab_input && ctx->world > 1) {
            int r0, r1;
            slab(H, ctx->rank, ctx->world, &r0, &r1);
            npx = (size_t)(r1 - r0) * W;
        }
        DevBuf<float> d_lum(std::max<size_t>(npx, 1));
        DevBuf<unsigned char> d_u8(std::max<size_t>(npx, 1));  // released after train_impl, which has drained the stream by then
        HIP_OK(hipMemcpyAsync(d_u8.p, h_lum8, npx, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(nlek::channel8_plane(ctx->stream, d_u8.p, (long long)npx, d_lum.p));
        nle_filter* f = train_impl(ctx, d_lum.p, H, W, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter, n_eigen_vectors);
        f->plane_bytes = d_lum.n * sizeof(float);
        f->d_plane = d_lum.take();
        *out = f;
    });
}

void nle_filter_destroy(nle_filter* f) {
    if (!f) return;
    if (f->ctx) f->ctx->filters.erase(f);
    if (f->d_V) arena_release(f->ctx, f->d_V, f->v_bytes);  // back to the ctx's workspace cache (or hipFree)
    if (f->d_plane) arena_release(f->ctx, f->d_plane, f->plane_bytes);
    if (f->d_V64) arena_release(f->ctx, f->d_V64, f->v64_bytes);
    for (auto& b : f->owned) arena_release(f->ctx, b.first, b.second);
    delete f;
}

int nle_filter_info(const nle_filter* f, long long* n_local, int* K, int* r, int* p, int* row0, int* row1) {
    if (!f) return NLE_ERR_INVALID;
    if (n_local) *n_local = f->n_local;
    if (K) *K = f->K;
    if (r) *r = f->r;
    if (p) *p = f->p;
    if (row0) *row0 = f->row0;
    if (row1) *row1 = f->row1;
    return NLE_OK;
}

int nle_filter_diag(const nle_filter* f, int* h_info) {
    if (!f || !h_info) return NLE_ERR_INVALID;
    const int v[8] = {f->formulation, f->p, f->r, f->r_wa, f->r_q, f->K, f->chol_ka, f->chol_wa};
    std::copy(v, v + 8, h_info);
    return NLE_OK;
}

int nle_filter_eigvals(const nle_filter* f, double* h_eigvals) {
    if (!f || !h_eigvals) return NLE_ERR_INVALID;
    std::copy(f->eigvals.begin(), f->eigvals.end(), h_eigvals);
    return NLE_OK;
}

int nle_filter_eigvec_range(const nle_filter* f, int ncols, double* h_min, double* h_max) {
    if (!f || !f->ctx || ncols < 1 || ncols > f->K || !h_min || !h_max) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        nle_ctx* c = f->ctx;
        HIP_OK(hipSetDevice(c->device));
        // an implicit V is not materialised for this: only the requested leading columns are projected, into a
        // temporary (the CLI prints the range of 5 of K columns, src/filter.cpp:506)
        const float* d_cols = f->d_V;
        int ldc = f->ldv;
        DevBuf<float> d_tmp;
        if (!f->d_V && f->lazy) {
            const long long M = f->n_local, pix0 = (long long)f->row0 * f->W;
            const int ldd_full = f->ldd, ldd = nlek::project64_ld(ncols);
            std::vector<double> Dfull((size_t)f->p * ldd_full), Dk((size_t)f->p * ldd, 0.0);
            HIP_OK(hipMemcpyAsync(Dfull.data(), f->d_D, Dfull.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
            for (int a = 0; a < f->p; ++a)
                for (int k = 0; k < ncols; ++k) Dk[(size_t)a * ldd + k] = Dfull[(size_t)a * ldd_full + k];
            DevBuf<double> d_Dk(Dk.size());
            HIP_OK(hipMemcpyAsync(d_Dk.p, Dk.data(), Dk.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            ldc = ld4(ncols);
            d_tmp.alloc((size_t)std::max<long long>(M, 1) * ldc);
            PROFILED(c, NLE_K_PROJECT, nlek::project64(c->stream, f->d_lum - pix0, f->gs, f->d_samples, f->p, f->nsw, f->npw,
                                                       pix0, M, d_Dk.p, ncols, f->d_c, d_tmp.p, ldc));
            scatter_sample_rows(c, f->h_sample_pix, f->p, f->h_Vrows, f->p, ncols, ldc, pix0, M, d_tmp.p);
            HIP_OK(hipStreamSynchronize(c->stream));  // Dk (host) is consumed
            d_cols = d_tmp.p;
        } else {
            ensure_V(const_cast<nle_filter*>(f));
            d_cols = f->d_V;
        }
        const int nb = 256;
        DevBuf<float> d_out((size_t)nb * 2 * ncols);
        HIP_OK(nlek::col_range(c->stream, d_cols, f->n_local, ldc, ncols, d_out.p, nb));
        std::vector<float> out((size_t)nb * 2 * ncols);
        HIP_OK(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < ncols; ++k) {
            double mn = out[2 * k], mx = out[2 * k + 1];
            for (int b = 1; b < nb; ++b) {
                mn = std::min(mn, (double)out[(size_t)b * 2 * ncols + 2 * k]);
                mx = std::max(mx, (double)out[(size_t)b * 2 * ncols + 2 * k + 1]);
            }
            h_min[k] = mn;
            h_max[k] = mx;
        }
    });
}

int nle_filter_eigvecs(const nle_filter* f, const float** d_V, int* ld) {
    if (!f || !f->ctx || !d_V || !ld) return NLE_ERR_INVALID;
    const int st = guard(f->ctx, [&] {
        HIP_OK(hipSetDevice(f->ctx->device));
        ensure_V(const_cast<nle_filter*>(f));
    });
    if (st != NLE_OK) return st;
    *d_V = f->d_V;
    *ld = f->ldv;
    return NLE_OK;
}

int nle_filter_copy_eigvecs(const nle_filter* f, float* d_out) {
    if (!f || !f->ctx || !d_out) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        HIP_OK(hipSetDevice(f->ctx->device));
        ensure_V(const_cast<nle_filter*>(f));
        HIP_OK(hipMemcpyAsync(d_out, f->d_V, (size_t)f->n_local * f->ldv * sizeof(float), hipMemcpyDeviceToDevice,
                              f->ctx->stream));
        HIP_OK(hipStreamSynchronize(f->ctx->stream));
    });
}

int nle_filter_timings(const nle_filter* f, double* h_ms) {
    if (!f || !h_ms) return NLE_ERR_INVALID;
    std::copy(f->ms, f->ms + 6, h_ms);
    return NLE_OK;
}

int nle_apply(nle_filter* f, const float* d_x, int H, int W, const double* h_fS, float* d_y) {
    if (!f || !f->ctx || !d_x || !h_fS || !d_y) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] { apply_impl(f, d_x, H, W, h_fS, 1, d_y); });
}

int nle_apply_layers(nle_filter* f, const float* d_x, int H, int W, int L, float* d_y) {
    if (!f || !f->ctx || !d_x || !d_y || L < 1) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        std::vector<double> resp((size_t)L * f->K);
        layer_resp(f->eigvals.data(), f->K, L, resp.data());
        apply_impl(f, d_x, H, W, resp.data(), L, d_y);
    });
}

static void apply_host_common(nle_filter* f, const float* h_x, int H, int W, const double* g, int L, float* h_y) {
    nle_ctx* c = f->ctx;
    if ((long long)H * W != (long long)f->H * f->W)
        throw Fail{NLE_ERR_INVALID, "Number of values in channel must match that of training image."};
    if (!h_x && !f->d_plane)
        throw Fail{NLE_ERR_INVALID, "h_x == NULL needs a filter trained by nle_train_host (it keeps the training plane)"};
    HIP_OK(hipSetDevice(c->device));
    DevBuf<float> d_xbuf, d_y((size_t)L * std::max<long long>(f->n_local, 1));
    const float* d_x = f->d_plane;
    if (h_x) {
        const size_t npx = (c->slab_input && c->world > 1) ? (size_t)f->n_local : (size_t)H * W;
        d_xbuf.alloc(std::max<size_t>(npx, 1));
        HIP_OK(hipMemcpyAsync(d_xbuf.p, h_x, npx * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_x = d_xbuf.p;
    }
    // each finished group of layers goes home on the copy stream while the next one is computed (the copies are only
    // asynchronous when h_y is pinned: nle_host_alloc)
    if (!c->copy_stream) {
        HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (auto& e : c->copy_ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    int flip = 0;
    const size_t n = (size_t)f->n_local;
    // whatever happens below (an exception out of apply_impl or of a later callback), no copy may still be reading d_y or
    // writing the caller's h_y when this function is left: d_y goes back to the ctx's cache in its destructor
    struct CopyDrain {
        nle_ctx* c;
        ~CopyDrain() {
            if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
            (void)hipStreamSynchronize(c->stream);
        }
    } drain{c};
    apply_impl(f, d_x, H, W, g, L, d_y.p, [&](int l0, int nl) {
        hipEvent_t ev = c->copy_ev[flip ^= 1];
        HIP_OK(hipEventRecord(ev, c->stream));
        HIP_OK(hipStreamWaitEvent(c->copy_stream, ev, 0));
        HIP_OK(hipMemcpyAsync(h_y + (size_t)l0 * n, d_y.p + (size_t)l0 * n, (size_t)nl * n * sizeof(float),
                              hipMemcpyDeviceToHost, c->copy_stream));
    }, 1);
    HIP_OK(hipStreamSynchronize(c->copy_stream));
    // d_y returns to the ctx's cache: order its next use on the main stream behind the copies
    HIP_OK(hipStreamSynchronize(c->stream));
}

int nle_apply_host(nle_filter* f, const float* h_x, int H, int W, const double* h_fS, float* h_y) {
    if (!f || !f->ctx || !h_fS || !h_y) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] { apply_host_common(f, h_x, H, W, h_fS, 1, h_y); });
}

int nle_apply_layers_host(nle_filter* f, const float* h_x, int H, int W, int L, float* h_y) {
    if (!f || !f->ctx || !h_y || L < 1) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        std::vector<double> resp((size_t)L * f->K);
        layer_resp(f->eigvals.data(), f->K, L, resp.data());
        apply_host_common(f, h_x, H, W, resp.data(), L, h_y);
    });
}

// the same plane kept as fp32 levels (what nle_lab2bgr8 / nle_lab2bgr8_planes take for a replaced channel)
int nle_apply_rounded8(nle_filter* f, const float* d_x, int H, int W, const double* h_fS, float* d_y) {
    if (!f || !f->ctx || !d_x || !h_fS || !d_y) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        apply_impl(f, d_x, H, W, h_fS, 1, d_y, nullptr, 0, /*round8=*/true);
        if (!(f->lazy && std::getenv("NLE_APPLY_WITH_V") == nullptr)) {  // formulations with fp32 planes: round those
            DevBuf<unsigned char> d_o((size_t)std::max<long long>(f->n_local, 1));
            HIP_OK(nlek::plane_to_u8(f->ctx->stream, d_y, f->n_local, d_o.p));
            HIP_OK(nlek::channel8_plane(f->ctx->stream, d_o.p, f->n_local, d_y));
            HIP_OK(hipStreamSynchronize(f->ctx->stream));
        }
    });
}

// NLEFilter::enhance's L plane (src/filter.cpp:428-436): apply, clamp, convertTo(CV_8U) -- one byte per pixel leaves the device
int nle_apply_u8(nle_filter* f, const float* d_x, int H, int W, const double* h_fS, unsigned char* d_out) {
    if (!f || !f->ctx || !d_x || !h_fS || !d_out) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        nle_ctx* c = f->ctx;
        DevBuf<float> d_y((size_t)std::max<long long>(f->n_local, 1));
        apply_impl(f, d_x, H, W, h_fS, 1, d_y.p, nullptr, 0, /*round8=*/true);
        HIP_OK(nlek::plane_to_u8(c->stream, d_y.p, f->n_local, d_out));
        HIP_OK(hipStreamSynchronize(c->stream));   // d_y returns to the ctx's cache
    });
}

int nle_apply_u8_host(nle_filter* f, const float* h_x, int H, int W, const double* h_fS, unsigned char* h_out) {
    if (!f || !f->ctx || !h_fS || !h_out) return NLE_ERR_INVALID;
    return guard(f->ctx, [&] {
        nle_ctx* c = f->ctx;
        if ((long long)H * W != (long long)f->H * f->W)
            throw Fail{NLE_ERR_INVALID, "Number of values in channel must match that of training image."};
        if (!h_x && !f->d_plane)
            throw Fail{NLE_ERR_INVALID, "h_x == NULL needs a filter trained by nle_train_host (it keeps the training plane)"};
        HIP_OK(hipSetDevice(c->device));
        const size_t n = (size_t)std::max<long long>(f->n_local, 1);
        DevBuf<float> d_xbuf, d_y(n);
        DevBuf<unsigned char> d_o(n);
        const float* d_x = f->d_plane;
        if (h_x) {
            const size_t npx = (c->slab_input && c->world > 1) ? (size_t)f->n_local : (size_t)H * W;
            d_xbuf.alloc(std::max<size_t>(npx, 1));
            HIP_OK(hipMemcpyAsync(d_xbuf.p, h_x, npx * sizeof(float), hipMemcpyHostToDevice, c->stream));
            d_x = d_xbuf.p;
        }
        struct Drain {   // no copy may still be writing the caller's buffer, nor a kernel using d_y, when this is left
            nle_ctx* c;
            ~Drain() { (void)hipStreamSynchronize(c->stream); }
        } drain{c};
        apply_impl(f, d_x, H, W, h_fS, 1, d_y.p, nullptr, 0, /*round8=*/true);
        HIP_OK(nlek::plane_to_u8(c->stream, d_y.p, f->n_local, d_o.p));
        HIP_OK(hipMemcpyAsync(h_out, d_o.p, (size_t)f->n_local, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    });
}

int nle_bench_affinity(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                       double hy, float* d_kab, int reps, double* h_avg_ms) {
    if (!ctx || !d_lum || !d_kab || reps < 1 || !h_avg_ms) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        check_image_size(H, W);
        GridSpec gs;
        if (!make_grid(H, W, n_row_samples, n_col_samples, &gs)) throw Fail{NLE_ERR_INVALID, "invalid sample counts"};
        HIP_OK(hipSetDevice(ctx->device));
        SampleSet ss = fetch_samples(ctx, d_lum, gs);
        int row0, row1;
        slab(H, ctx->rank, ctx->world, &row0, &row1);
        DevBuf<float4> d_samples(ss.p);
        HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), ss.p * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        const float sw = nsw_of(hx), pw = nsw_of(hy);
        const long long pix0 = (long long)row0 * W, M = (long long)(row1 - row0) * W;
        HIP_OK(nlek::affinity(ctx->stream, d_lum, gs, d_samples.p, ss.p, ld4(ss.p), sw, pw, pix0, M, d_kab));
        Timer tm(ctx->stream);
        tm.start();
        for (int i = 0; i < reps; ++i)
            HIP_OK(nlek::affinity(ctx->stream, d_lum, gs, d_samples.p, ss.p, ld4(ss.p), sw, pw, pix0, M, d_kab));
        tm.stop();
        *h_avg_ms = tm.ms() / reps;
    });
}

int nle_bench_affinity64(nle_ctx* ctx, const float* d_lum, int H, int W, int n_row_samples, int n_col_samples, double hx,
                         double hy, long long rows, double* d_kab, int reps, double* h_avg_ms) {
    if (!ctx || !d_lum || !d_kab || reps < 1 || !h_avg_ms || rows < 1) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        check_image_size(H, W);
        GridSpec gs;
        if (!make_grid(H, W, n_row_samples, n_col_samples, &gs)) throw Fail{NLE_ERR_INVALID, "invalid sample counts"};
        HIP_OK(hipSetDevice(ctx->device));
        SampleSet ss = fetch_samples(ctx, d_lum, gs);
        int row0, row1;
        slab(H, ctx->rank, ctx->world, &row0, &row1);
        DevBuf<float4> d_samples(ss.p);
        HIP_OK(hipMemcpyAsync(d_samples.p, ss.packed.data(), ss.p * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        const long long pix0 = (long long)row0 * W, M = std::min<long long>(rows, row1 - row0) * W;
        const double sw = 1.0 / (hx * hx), pw = 1.0 / (hy * hy);
        HIP_OK(nlek::affinity64(ctx->stream, d_lum, gs, d_samples.p, ss.p, ld4(ss.p), sw, pw, pix0, M, d_kab, true));
        Timer tm(ctx->stream);
        tm.start();
        for (int i = 0; i < reps; ++i)
            HIP_OK(nlek::affinity64(ctx->stream, d_lum, gs, d_samples.p, ss.p, ld4(ss.p), sw, pw, pix0, M, d_kab, true));
        tm.stop();
        *h_avg_ms = tm.ms() / reps;
    });
}

int nle_filter_level_tiles(const nle_filter* f, int* first_tile, int* n_tiles) {
    if (!f || !first_tile || !n_tiles) return NLE_ERR_INVALID;
    *first_tile = f->has_sorted ? f->sorted.lev_t0 : 0;
    *n_tiles = f->has_sorted ? f->sorted.lev_nt : 16;
    return NLE_OK;
}

int nle_bench_sinkhorn_pass(nle_ctx* ctx, const float* d_phi, long long M, int ld, int r, int reps, double* h_avg_ms) {
    if (!ctx || !d_phi || reps < 1 || !h_avg_ms || M < 1 || r < 1 || ld < r || (ld & 3)) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        std::vector<double> ones(ld, 1.0);
        DevBuf<double> d_lam(ld), d_t(ld), d_partial((size_t)nlek::kRowpassMaxBlocks * ld);
        HIP_OK(hipMemcpyAsync(d_lam.p, ones.data(), ld * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemcpyAsync(d_t.p, ones.data(), ld * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        int nb = 0;
        HIP_OK(nlek::rowpass(ctx->stream, nlek::ROWPASS_RECIP, d_phi, M, ld, d_t.p, d_lam.p, nullptr, NLE_EPS, d_partial.p, &nb));
        Timer tm(ctx->stream);
        tm.start();
        for (int i = 0; i < reps; ++i)
            HIP_OK(nlek::rowpass(ctx->stream, nlek::ROWPASS_RECIP, d_phi, M, ld, d_t.p, d_lam.p, nullptr, NLE_EPS, d_partial.p, &nb));
        tm.stop();
        *h_avg_ms = tm.ms() / reps;
    });
}

}  // extern "C"
