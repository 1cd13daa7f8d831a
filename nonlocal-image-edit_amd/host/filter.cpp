// Host side of the C++ drop-in surface (include/nle/filter.hpp) over the C ABI (include/nle.h).
// Mirrors the reference's src/filter.cpp function by function; every N-sized product runs in
// libnle_hip.so on the GPU (fp32 storage, fp64 reductions), the p x p algebra stays here in fp64.
#include "nle/filter.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <algorithm>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <string>
#include <thread>

#include "nle.h"

namespace nle {

namespace {

// one context per process, created on first use (the reference keeps no global state; this one
// only holds the HIP stream)
nle_ctx* shared_ctx() {
    static nle_ctx* ctx = nullptr;
    if (!ctx) {
        int dev = 0;
        if (const char* e = std::getenv("NLE_DEVICE")) dev = std::atoi(e);
        if (nle_ctx_create(dev, nullptr, &ctx) != NLE_OK)
            throw std::runtime_error(std::string("nle: cannot create GPU context: ") + nle_last_error(nullptr));
        if (const char* m = std::getenv("NLE_MODE")) nle_ctx_set_mode(ctx, std::atoi(m));
    }
    return ctx;
}

void check(int status, nle_ctx* ctx) {
    if (status != NLE_OK) throw std::runtime_error(nle_last_error(ctx));
}

// ---- NLE_DEVICES=<dev>,<dev>,...  (SURVEY.md section 8e through the reference's own surface) ----
// One context per listed device, created once per process; rank r owns image rows nle_slab_rows(H, r, G) and hands the
// library only those (nle_ctx_set_slab_input).  All-reduces: the library's own RCCL communicator when the devices are
// distinct (ncclCommInitRank from G threads), otherwise -- the same device listed more than once, a rehearsal on one
// GPU -- a host-mediated sum in rank order behind a thread barrier, so that every rank sees bitwise the same sums.
struct DeviceGroup {
    std::vector<int> devs;
    std::vector<nle_ctx*> ctx;
    bool native = false;
    // host-mediated all-reduce
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long long generation = 0;
    bool failed = false;
    int bound_p = -1;           // sample count the communicator / comm buffers are bound for (-1: not bound)
    bool native_bound = false;  // the library's RCCL communicator exists on every ctx
    std::vector<std::vector<double>> slot;  // one per rank
    std::vector<void*> d_comm;
    struct Rank { DeviceGroup* g; int r; };
    std::vector<Rank> ranks;

    int size() const { return (int)devs.size(); }
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) throw std::runtime_error("nle: another rank of the device group failed");
        const long long gen = generation;
        if (++arrived == size()) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
            if (failed) throw std::runtime_error("nle: another rank of the device group failed");
        }
    }
    void fail() {
        {
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
            cv.notify_all();  // ranks waiting in the host-mediated all-reduce
        }
        // ranks waiting in a native collective for the one that failed: abort every communicator of the group, their
        // pending ncclAllReduce ends and their next call returns NLE_ERR_COMM; device_group() rebuilds the group (a fresh
        // unique id, ncclCommInitRank on every ctx) before the next train
        if (native)
            for (nle_ctx* c : ctx)
                if (c) (void)nle_ctx_abort_rccl(c);
    }
    // after a failed run (every rank thread has been joined): forget the failure and whatever was bound, so that the next
    // train binds a fresh communicator / fresh comm buffers instead of failing for the rest of the process
    void recover() {
        std::lock_guard<std::mutex> lk(mu);
        failed = false;
        arrived = 0;
        bound_p = -1;
        native_bound = false;
    }
    static int allreduce_cb(void* user, void* d_buf, size_t count) {
        Rank* me = static_cast<Rank*>(user);
        DeviceGroup* g = me->g;
        try {
            std::vector<double>& mine = g->slot[me->r];
            mine.resize(count);
            if (nle_dev_download(g->ctx[me->r], mine.data(), d_buf, count * sizeof(double)) != NLE_OK) return 1;
            g->barrier();  // every slot is filled
            std::vector<double> sum(count, 0.0);
            for (int r = 0; r < g->size(); ++r)  // rank order: identical on every rank
                for (size_t i = 0; i < count; ++i) sum[i] += g->slot[r][i];
            g->barrier();  // every rank has read every slot
            return nle_dev_upload(g->ctx[me->r], d_buf, sum.data(), count * sizeof(double)) == NLE_OK ? 0 : 1;
        } catch (...) {
            return 1;
        }
    }
    // fn(rank) on one thread per rank; the first exception is rethrown after all threads have ended
    template <typename Fn>
    void run(Fn&& fn) {
        std::vector<std::thread> th;
        std::vector<std::exception_ptr> err(size());
        for (int r = 0; r < size(); ++r)
            th.emplace_back([&, r] {
                try {
                    fn(r);
                } catch (...) {
                    err[r] = std::current_exception();
                    fail();
                }
            });
        for (auto& t : th) t.join();
        for (auto& e : err)
            if (e) std::rethrow_exception(e);
    }
};

// nullptr unless NLE_DEVICES lists at least two devices
DeviceGroup* device_group(int p_samples) {
    static DeviceGroup* g = nullptr;
    static bool looked = false;
    if (!looked) {
        looked = true;
        const char* e = std::getenv("NLE_DEVICES");
        std::vector<int> devs;
        if (e) {
            std::string s(e);
            size_t pos = 0;
            while (pos <= s.size()) {
                const size_t c = s.find(',', pos);
                const std::string tok = s.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
                if (!tok.empty()) devs.push_back(std::stoi(tok));
                if (c == std::string::npos) break;
                pos = c + 1;
            }
        }
        if (devs.size() >= 2) {
            g = new DeviceGroup;
            g->devs = devs;
            std::vector<int> sorted(devs);
            std::sort(sorted.begin(), sorted.end());
            g->native = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
            const int G = g->size();
            g->ctx.resize(G, nullptr);
            g->slot.resize(G);
            g->d_comm.resize(G, nullptr);
            g->ranks.resize(G);
            for (int r = 0; r < G; ++r) {
                if (nle_ctx_create(devs[r], nullptr, &g->ctx[r]) != NLE_OK)
                    throw std::runtime_error(std::string("nle: cannot create GPU context: ") + nle_last_error(nullptr));
                if (const char* m = std::getenv("NLE_MODE")) nle_ctx_set_mode(g->ctx[r], std::atoi(m));
                g->ranks[r] = DeviceGroup::Rank{g, r};
            }
        }
    }
    if (g && p_samples > 0) {
        if (g->failed) g->recover();  // an earlier train / enhance of this process failed on some rank
        // (re)bind the communicator for this sample count: the callback form needs a comm buffer of nle_comm_len(p)
        if (g->bound_p != p_samples) {
            const int G = g->size();
            if (g->native) {
                if (!g->native_bound) {
                    unsigned char id[NLE_RCCL_UNIQUE_ID_BYTES];
                    check(nle_rccl_unique_id(id, sizeof id), nullptr);
                    g->run([&](int r) { check(nle_ctx_init_rccl(g->ctx[r], r, G, id, sizeof id), g->ctx[r]); });
                    g->native_bound = true;
                }
            } else {
                const size_t len = nle_comm_len(p_samples);
                for (int r = 0; r < G; ++r) {
                    if (g->d_comm[r]) nle_dev_free(g->ctx[r], g->d_comm[r]);
                    check(nle_dev_alloc(g->ctx[r], len * sizeof(double), &g->d_comm[r]), g->ctx[r]);
                    check(nle_ctx_set_shard(g->ctx[r], r, G, &DeviceGroup::allreduce_cb, &g->ranks[r],
                                            static_cast<double*>(g->d_comm[r]), len), g->ctx[r]);
                }
            }
            for (int r = 0; r < G; ++r) check(nle_ctx_set_slab_input(g->ctx[r], 1), g->ctx[r]);
            g->bound_p = p_samples;
        }
    }
    return g;
}

// RAII device buffer through the ABI helpers
struct Dev {
    nle_ctx* c;
    void* p = nullptr;
    Dev(nle_ctx* ctx, size_t bytes) : c(ctx) { check(nle_dev_alloc(c, bytes, &p), c); }
    ~Dev() { nle_dev_free(c, p); }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    float* f() { return static_cast<float*>(p); }
    double* d() { return static_cast<double*>(p); }
};

inline double recip0(double v, double eps = EPS) { return std::fabs(v) >= eps ? 1.0 / v : 0.0; }  // :42-54

std::vector<float> plane_f32(const Image& m) {
    if (m.channels() != 1 || m.depth() != NLE_64F) throw std::runtime_error("expected a 1-channel CV_64F matrix");
    std::vector<float> v(m.total());
    const double* s = m.ptr<double>();
    for (size_t i = 0; i < v.size(); ++i) v[i] = (float)s[i];
    return v;
}

// rows of a column-major p x n fp64 matrix's TRANSPOSE as fp32 row-per-pixel (n x ld), i.e. the
// reference's `Kab` / `Wab` seen as one row per pixel
std::vector<float> cols_as_rows_f32(const Mat& m, int ld) {
    std::vector<float> v((size_t)m.cols() * ld, 0.f);
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) v[(size_t)j * ld + i] = (float)m(i, j);
    return v;
}

// the same in fp64 (the stage-level API runs on fp64 device matrices: nle_*64)
std::vector<double> cols_as_rows_f64(const Mat& m, int ld) {
    std::vector<double> v((size_t)m.cols() * ld, 0.0);
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) v[(size_t)j * ld + i] = m(i, j);
    return v;
}

}  // namespace

Mat operator*(const Mat& a, const Mat& b) {
    if (a.cols() != b.rows()) throw std::runtime_error("matrix product: shape mismatch");
    Mat c(a.rows(), b.cols());
    for (int j = 0; j < b.cols(); ++j)
        for (int k = 0; k < a.cols(); ++k) {
            const double bv = b(k, j);
            for (int i = 0; i < a.rows(); ++i) c(i, j) += a(i, k) * bv;
        }
    return c;
}

Image eigen2opencv(const Vec& v, int nrows, int ncols) {  // include/utils.hpp:21-26 (clone)
    Image m(nrows, ncols, NLE_64F, 1);
    std::copy(v.data(), v.data() + (size_t)nrows * ncols, m.ptr<double>());
    return m;
}

Vec opencv2eigen(const Image& mat) {  // include/utils.hpp:28-41, row-major flatten
    Vec lv((int)mat.total());
    int k = 0;
    for (int i = 0; i < mat.rows; i++)
        for (int j = 0; j < mat.cols; j++) lv(k++) = mat.at<double>(i, j);
    return lv;
}

// ------------------------------------------------------------------ computeKernel, :114-167
std::tuple<Permutation, Mat, Mat> computeKernel(const Image& mat, int nRowSamples, int nColSamples, DType hx,
                                                DType hy) {
    if (nRowSamples > mat.rows || nColSamples > mat.cols)
        throw std::runtime_error("Number of samples per row and col must be <= that of image.");
    nle_ctx* c = shared_ctx();
    const int H = mat.rows, W = mat.cols;
    const long long N = (long long)H * W;
    int rs, ro, nr, cs, co, nc;
    if (nle_sample_grid(H, W, nRowSamples, nColSamples, &rs, &ro, &nr, &cs, &co, &nc) != NLE_OK)
        throw std::runtime_error("Number of samples per row and col must be <= that of image.");
    const int p = nr * nc, ld = nle_ld(p);
    std::vector<float> lum = plane_f32(mat);
    Dev d_lum(c, (size_t)N * 4), d_kab(c, (size_t)N * ld * 8);
    check(nle_dev_upload(c, d_lum.p, lum.data(), (size_t)N * 4), c);
    Mat Ka(p, p);
    check(nle_compute_kernel64(c, d_lum.f(), H, W, nRowSamples, nColSamples, hx, hy, Ka.data(), d_kab.d()), c);
    std::vector<double> kab((size_t)N * ld);
    check(nle_dev_download(c, kab.data(), d_kab.p, kab.size() * 8), c);
    // [selected; rest] order, both in row-major scan order (:56-80, :156-164)
    Permutation P;
    P.idx.resize((size_t)N);
    std::vector<char> sel((size_t)N, 0);
    int k = 0;
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) {
            const int pix = to1DIndex(ro + i * rs, co + j * cs, W);
            P.idx[(size_t)k++] = pix;
            sel[(size_t)pix] = 1;
        }
    Mat Kab(p, (int)(N - p));
    int jrest = 0;
    for (long long pix = 0; pix < N; ++pix) {
        if (sel[(size_t)pix]) continue;
        P.idx[(size_t)(p + jrest)] = (int)pix;
        const double* row = kab.data() + (size_t)pix * ld;
        for (int i = 0; i < p; ++i) Kab(i, jrest) = row[i];
        ++jrest;
    }
    return std::make_tuple(P, Ka, Kab);
}

// ------------------------------------------------------------------ eigenDecomposition, :204-228
std::pair<Mat, Vec> eigenDecomposition(const Mat& M, DType eps) {
    const int n = M.rows();
    if (n == 0 || M.cols() != n) throw std::runtime_error("eigenDecomposition: square matrix expected");
    Mat U(n, n);
    Vec D(n);
    int r = 0;
    if (nle_eigen_decomposition(M.data(), n, eps, U.data(), D.data(), &r) != NLE_OK)
        throw std::runtime_error("eigenDecomposition: no convergence");
    return std::make_pair(U.leftCols(r), D.head(r));
}

// ------------------------------------------------------------------ nystromApproximation, :257-280
std::pair<Vec, Mat> nystromApproximation(const Mat& Ka, const Mat& Kab) {
    nle_ctx* c = shared_ctx();
    Mat eigvecs;
    Vec eigvals;
    std::tie(eigvecs, eigvals) = eigenDecomposition(Ka);
    int nnz = 0;
    for (int i = 0; i < eigvals.size(); ++i)
        if (std::fabs(eigvals(i)) >= EPS) ++nnz;  // :265-266
    eigvecs = eigvecs.leftCols(nnz);
    eigvals = eigvals.head(nnz);
    const int p = Ka.rows(), r = nnz, n_rest = Kab.cols(), n = p + n_rest;
    Mat B(p, r);  // eigvecs * invEigVals
    for (int k = 0; k < r; ++k)
        for (int s = 0; s < p; ++s) B(s, k) = eigvecs(s, k) * recip0(eigvals(k));
    Mat phi(n, r);
    for (int k = 0; k < r; ++k)
        for (int s = 0; s < p; ++s) phi(s, k) = eigvecs(s, k);
    if (n_rest > 0 && r > 0) {
        const int lda = nle_ld(p), ldc = nle_ld(r);
        std::vector<double> rows = cols_as_rows_f64(Kab, lda);
        Dev d_A(c, rows.size() * 8), d_C(c, (size_t)n_rest * ldc * 8);
        check(nle_dev_upload(c, d_A.p, rows.data(), rows.size() * 8), c);
        check(nle_ts_gemm64(c, d_A.d(), n_rest, lda, p, B.data(), r, d_C.d()), c);  // Kab^T * B, :275
        std::vector<double> out((size_t)n_rest * ldc);
        check(nle_dev_download(c, out.data(), d_C.p, out.size() * 8), c);
        for (int j = 0; j < n_rest; ++j)
            for (int k = 0; k < r; ++k) phi(p + j, k) = out[(size_t)j * ldc + k];
    }
    return std::make_pair(eigvals, phi);
}

// ------------------------------------------------------------------ sinkhorn, :230-254
std::pair<Mat, Mat> sinkhorn(const Mat& phi, const Vec& eigvals, int maxIter) {
    nle_ctx* c = shared_ctx();
    const int n = phi.rows(), r = phi.cols();
    if (r == 0 || n == 0) throw std::runtime_error("sinkhorn: empty phi");
    if (maxIter < 1) throw std::runtime_error("sinkhorn: maxIter must be >= 1");
    const int ld = nle_ld(r);
    std::vector<double> rows((size_t)n * ld, 0.0);
    for (int k = 0; k < r; ++k)
        for (int i = 0; i < n; ++i) rows[(size_t)i * ld + k] = phi(i, k);
    Dev d_phi(c, rows.size() * 8), d_c(c, (size_t)n * 8);
    check(nle_dev_upload(c, d_phi.p, rows.data(), rows.size() * 8), c);
    std::vector<double> u_c(r), u_r(r);
    check(nle_sinkhorn_scalings64(c, d_phi.d(), n, ld, r, eigvals.data(), maxIter, u_c.data(), u_r.data()), c);
    check(nle_row_scalings64(c, d_phi.d(), n, ld, r, u_c.data(), d_c.d()), c);
    std::vector<double> cv((size_t)n);
    check(nle_dev_download(c, cv.data(), d_c.p, cv.size() * 8), c);
    const int q = r;  // :247  p = phi.cols()
    if (q > n) throw std::runtime_error("sinkhorn: phi has more columns than rows");
    // left = R * (phi_top * D)
    Mat left(q, r), right(q, r);
    for (int a = 0; a < q; ++a) {
        double sr = 0.0;
        for (int k = 0; k < r; ++k) sr += rows[(size_t)a * ld + k] * u_r[k];
        const double ra = recip0(sr);
        for (int k = 0; k < r; ++k) {
            const double v = rows[(size_t)a * ld + k];
            left(a, k) = ra * v * eigvals(k);
            right(a, k) = cv[(size_t)a] * v;
        }
    }
    Mat Wa = left * right.transpose();  // :249
    Mat Wab(q, n - q);
    if (n > q) {
        const int ldq = nle_ld(q);
        Mat lt = left.transpose();  // r x q
        Dev d_out(c, (size_t)(n - q) * ldq * 8);
        check(nle_ts_gemm64(c, d_phi.d() + (size_t)q * ld, n - q, ld, r, lt.data(), q, d_out.d()), c);
        std::vector<double> out((size_t)(n - q) * ldq);
        check(nle_dev_download(c, out.data(), d_out.p, out.size() * 8), c);
        for (int j = 0; j < n - q; ++j)
            for (int a = 0; a < q; ++a) Wab(a, j) = out[(size_t)j * ldq + a] * cv[(size_t)(q + j)];  // :250
    }
    return std::make_pair(Wa, Wab);
}

// ------------------------------------------------------------------ orthogonalize, :282-331
std::pair<Mat, Vec> orthogonalize(const Mat& Wa, const Mat& Wab, int nEigVectors, DType eps) {
    nle_ctx* c = shared_ctx();
    const int q = Wa.rows(), nb = Wab.cols();
    Mat eigvecs;
    Vec eigvals;
    std::tie(eigvecs, eigvals) = eigenDecomposition(Wa, eps);
    const int r2 = eigvals.size();
    Mat Us(q, r2);
    for (int k = 0; k < r2; ++k) {
        const double s = std::sqrt(recip0(eigvals(k), eps));
        for (int i = 0; i < q; ++i) Us(i, k) = eigvecs(i, k) * s;
    }
    Mat invRootWa = Us * eigvecs.transpose();  // :292
    Mat G(q, q);
    const int ldq = nle_ld(q);
    std::vector<double> rows = cols_as_rows_f64(Wab, ldq);  // Wab^T, one row per pixel
    Dev d_X(c, std::max<size_t>(rows.size(), 1) * 8);
    if (nb > 0) {
        check(nle_dev_upload(c, d_X.p, rows.data(), rows.size() * 8), c);
        check(nle_gram64(c, d_X.d(), nb, ldq, q, nullptr, G.data()), c);  // Wab * Wab^T, :296
    }
    Mat Q = invRootWa * G * invRootWa;
    for (int j = 0; j < q; ++j)
        for (int i = 0; i < q; ++i) Q(i, j) += Wa(i, j);
    Mat Vq;
    Vec Sq;
    std::tie(Vq, Sq) = eigenDecomposition(Q, eps);  // :313
    const int k = std::min(nEigVectors, Vq.cols());
    Vq = Vq.leftCols(k);
    Sq = Sq.head(k);
    Mat C = invRootWa * Vq;  // q x k
    for (int j = 0; j < k; ++j) {
        const double s = std::sqrt(recip0(Sq(j), eps));  // :319-321
        for (int i = 0; i < q; ++i) C(i, j) *= s;
    }
    Mat V(q + nb, k);
    Mat top = Wa * C;
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < q; ++i) V(i, j) = top(i, j);
    if (nb > 0 && k > 0) {
        const int ldk = nle_ld(k);
        Dev d_out(c, (size_t)nb * ldk * 8);
        check(nle_ts_gemm64(c, d_X.d(), nb, ldq, q, C.data(), k, d_out.d()), c);  // Wab^T * C, :327
        std::vector<double> out((size_t)nb * ldk);
        check(nle_dev_download(c, out.data(), d_out.p, out.size() * 8), c);
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j < k; ++j) V(q + i, j) = out[(size_t)i * ldk + j];
    }
    return std::make_pair(V, Sq);
}

// ------------------------------------------------------------------ transformEigenValues, :334-347
Vec transformEigenValues(const Vec& eigvals, const std::vector<DType>& weights) {
    if (weights.empty()) throw std::runtime_error("transformEigenValues: at least one weight is required");
    Vec fS(eigvals.size());
    if (nle_transform_eigenvalues(eigvals.data(), eigvals.size(), weights.data(), (int)weights.size(), fS.data()) !=
        NLE_OK)
        throw std::runtime_error("transformEigenValues: bad arguments");
    return fS;
}

// ------------------------------------------------------------------ colour wrapper (host)
// cv::cvtColor on 8-bit images: both directions are OpenCV's integer table algorithms (tables from the C ABI)
Image bgr2lab8(const Image& bgr) {
    if (bgr.channels() != 3 || bgr.depth() != NLE_8U) throw std::runtime_error("bgr2lab8: 8UC3 image expected");
    // OpenCV's fixed-point 8-bit path (imgproc RGB2Lab_b; include/nle.h at nle_lab8_tables): exact integers
    unsigned short gam[256], cb[3072];
    int k[9];
    if (nle_lab8_tables(gam, cb, k) != NLE_OK) throw std::runtime_error("bgr2lab8: tables");
    auto sat = [](int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    Image lab(bgr.rows, bgr.cols, NLE_8U, 3);
    const unsigned char* s = bgr.ptr<unsigned char>();
    unsigned char* d = lab.ptr<unsigned char>();
    for (size_t i = 0; i < bgr.total(); ++i, s += 3, d += 3) {
        const int B = gam[s[0]], G = gam[s[1]], R = gam[s[2]];
        const int fX = cb[(R * k[0] + G * k[1] + B * k[2] + 2048) >> 12];
        const int fY = cb[(R * k[3] + G * k[4] + B * k[5] + 2048) >> 12];
        const int fZ = cb[(R * k[6] + G * k[7] + B * k[8] + 2048) >> 12];
        d[0] = sat((296 * fY - 1336934 + 16384) >> 15);
        d[1] = sat((500 * (fX - fY) + 128 * 32768 + 16384) >> 15);
        d[2] = sat((200 * (fY - fZ) + 128 * 32768 + 16384) >> 15);
    }
    return lab;
}

Image lab2bgr8(const Image& lab) {
    if (lab.channels() != 3 || lab.depth() != NLE_8U) throw std::runtime_error("lab2bgr8: 8UC3 image expected");
    // OpenCV's integer 8-bit path (imgproc Lab2RGBinteger; include/nle.h at nle_lab8_inverse_tables): exact integers
    static std::vector<int> ab(36864);
    static unsigned short yf[512], ig[4096];
    static int k[9];
    static const bool ok = nle_lab8_inverse_tables(yf, ab.data(), ig, k) == NLE_OK;
    if (!ok) throw std::runtime_error("lab2bgr8: tables");
    Image bgr(lab.rows, lab.cols, NLE_8U, 3);
    const unsigned char* s = lab.ptr<unsigned char>();
    unsigned char* d = bgr.ptr<unsigned char>();
    auto enc = [&](const int* c, int x, int y, int z) {
        const int v = (c[0] * x + c[1] * y + c[2] * z + (1 << 13)) >> 14;
        return (unsigned char)ig[std::min(4095, std::max(0, v))];
    };
    for (size_t i = 0; i < lab.total(); ++i, s += 3, d += 3) {
        const int y = yf[2 * s[0]], fy = yf[2 * s[0] + 1];
        const int x = ab[fy + (((5 * s[1] * 53687 + (1 << 7)) >> 13) - 4194) + 8145];
        const int z = ab[fy - (((s[2] * 41943 + (1 << 4)) >> 9) - 10484) + 8145];
        d[0] = enc(k + 6, x, y, z);
        d[1] = enc(k + 3, x, y, z);
        d[2] = enc(k, x, y, z);
    }
    return bgr;
}

// the same conversions on the GPU (csrc/colour.hip) -- what NLEFilter uses; bit-identical to the host functions above
Image bgr2lab8_device(const Image& bgr) {
    if (bgr.channels() != 3 || bgr.depth() != NLE_8U) throw std::runtime_error("bgr2lab8: 8UC3 image expected");
    nle_ctx* c = shared_ctx();
    const size_t n = bgr.total();
    Dev d_in(c, n * 3), d_out(c, n * 3);
    check(nle_dev_upload(c, d_in.p, bgr.ptr<unsigned char>(), n * 3), c);
    check(nle_bgr2lab8(c, static_cast<unsigned char*>(d_in.p), (long long)n, static_cast<unsigned char*>(d_out.p), nullptr), c);
    Image lab(bgr.rows, bgr.cols, NLE_8U, 3);
    check(nle_dev_download(c, lab.ptr<unsigned char>(), d_out.p, n * 3), c);
    return lab;
}

Image lab2bgr8_device(const Image& lab) {
    if (lab.channels() != 3 || lab.depth() != NLE_8U) throw std::runtime_error("lab2bgr8: 8UC3 image expected");
    nle_ctx* c = shared_ctx();
    const size_t n = lab.total();
    Dev d_in(c, n * 3), d_out(c, n * 3);
    check(nle_dev_upload(c, d_in.p, lab.ptr<unsigned char>(), n * 3), c);
    check(nle_lab2bgr8(c, static_cast<unsigned char*>(d_in.p), nullptr, (long long)n, static_cast<unsigned char*>(d_out.p)), c);
    Image bgr(lab.rows, lab.cols, NLE_8U, 3);
    check(nle_dev_download(c, bgr.ptr<unsigned char>(), d_out.p, n * 3), c);
    return bgr;
}

namespace {
Image luminance_plane(const Image& lab) {  // split + convertTo(CV_64F), :460-469
    Image L(lab.rows, lab.cols, NLE_64F, 1);
    const unsigned char* s = lab.ptr<unsigned char>();
    double* d = L.ptr<double>();
    for (size_t i = 0; i < lab.total(); ++i) d[i] = s[3 * i];
    return L;
}
}  // namespace

// ------------------------------------------------------------------ NLEFilter
NLEFilter::NLEFilter() = default;
NLEFilter::~NLEFilter() = default;

void NLEFilter::trainFilter(const Image& channel, int nRowSamples, int nColSamples, DType hx, DType hy,
                            int nSinkhornIter, int nEigenVectors) {  // :480-512
    if (nRowSamples > channel.rows || nColSamples > channel.cols)
        throw std::runtime_error("Number of samples per row and col must be <= that of image.");
    ctx_ = shared_ctx();
    std::vector<float> lum = plane_f32(channel);
    Dev d_lum(ctx_, lum.size() * 4);
    check(nle_dev_upload(ctx_, d_lum.p, lum.data(), lum.size() * 4), ctx_);
    trainOnDevice(d_lum.f(), channel.rows, channel.cols, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors);
}

void NLEFilter::trainOnDevice(const float* d_lum, int rows, int cols, int nRowSamples, int nColSamples, DType hx,
                              DType hy, int nSinkhornIter, int nEigenVectors) {
    ctx_ = shared_ctx();
    fh_.reset();
    f_ = nullptr;
    group_.clear();
    if (verbose) {
        // the four stages run as one fused GPU pipeline; the banners keep the reference's stdout (:483-498)
        std::cout << "Computing kernel" << std::endl;
        std::cout << "Nystrom approximation" << std::endl;
        std::cout << "Sinkhorn" << std::endl;
        std::cout << "Orthogonalize" << std::endl;
    }
    check(nle_train(ctx_, d_lum, rows, cols, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors, &f_), ctx_);
    fh_.reset(f_, [](nle_filter* f) { nle_filter_destroy(f); });
    rows_ = rows;
    cols_ = cols;
    if (verbose) {
        Vec ev = eigvals();
        const int nshow = std::min(std::min(nEigenVectors, 5), ev.size());  // :504-506 (imshow dropped)
        double mn[5], mx[5];
        if (nshow > 0) check(nle_filter_eigvec_range(f_, nshow, mn, mx), ctx_);  // min/max on the device
        for (int i = 0; i < nshow; i++)
            std::cout << "Eigvec " << i << " eigval: " << ev(i) << " minCoeff: " << mn[i] << " maxCoeff: " << mx[i]
                      << std::endl;
    }
}

void NLEFilter::trainForEnhancement(const Image& image, int nRowSamples, int nColSamples, DType hx, DType hy,
                                    int nSinkhornIter, int nEigenVectors) {  // :514-519
    if (image.channels() != 3 || image.depth() != NLE_8U) throw std::runtime_error("Can only enhance RGB image.");
    if (nRowSamples > image.rows || nColSamples > image.cols)
        throw std::runtime_error("Number of samples per row and col must be <= that of image.");
    if (std::getenv("NLE_DEVICES") != nullptr) {
        int rs, ro, nr, cs, co, nc;
        if (nle_sample_grid(image.rows, image.cols, nRowSamples, nColSamples, &rs, &ro, &nr, &cs, &co, &nc) == NLE_OK &&
            device_group(nr * nc) != nullptr) {
            trainForEnhancementGroup(image, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors);
            return;
        }
    }
    // getLuminanceChannel (:460-469) on the device: BGR -> Lab (8 bit) -> L as float
    ctx_ = shared_ctx();
    const size_t n = image.total();
    Dev d_bgr(ctx_, n * 3), d_L(ctx_, n * 4);
    check(nle_dev_upload(ctx_, d_bgr.p, image.ptr<unsigned char>(), n * 3), ctx_);
    check(nle_bgr2lab8(ctx_, static_cast<unsigned char*>(d_bgr.p), (long long)n, nullptr, d_L.f()), ctx_);
    trainOnDevice(d_L.f(), image.rows, image.cols, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors);
}

// ---- denoise wrapper (src/filter.cpp:349-410, 521-538) ----
namespace {
int reflect101(int i, int n) {  // cv::BORDER_DEFAULT
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}
}  // namespace

Image bilateralFilter8(const Image& plane, double sigmaColor, double sigmaSpace) {
    if (plane.channels() != 1 || plane.depth() != NLE_8U) throw std::runtime_error("bilateralFilter8: 8UC1 image expected");
    int radius = 0;
    nle_bilateral_tables(sigmaColor, sigmaSpace, &radius, nullptr, nullptr);
    const int d = 2 * radius + 1, H = plane.rows, W = plane.cols;
    std::vector<float> sw((size_t)d * d), cw(256);
    nle_bilateral_tables(sigmaColor, sigmaSpace, &radius, sw.data(), cw.data());
    Image out(H, W, NLE_8U, 1);
    std::vector<int> ry(d), rx(d);
    for (int y = 0; y < H; ++y) {
        for (int i = 0; i < d; ++i) ry[i] = reflect101(y - radius + i, H);
        for (int x = 0; x < W; ++x) {
            for (int j = 0; j < d; ++j) rx[j] = reflect101(x - radius + j, W);
            const float v0 = (float)plane.at<unsigned char>(y, x);
            volatile float sum = 0.f, wsum = 0.f;  // volatile: one rounding per operation, like the device kernel
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    const float s = sw[(size_t)i * d + j];
                    if (s == 0.f) continue;
                    const float v = (float)plane.at<unsigned char>(ry[i], rx[j]);
                    volatile float w = s * cw[(int)std::fabs(v - v0)];
                    volatile float vw = v * w;
                    sum = sum + vw;
                    wsum = wsum + w;
                }
            volatile float q = sum / wsum;
            out.at<unsigned char>(y, x) = (unsigned char)std::nearbyint((float)q);
        }
    }
    return out;
}

Image bilateralFilter8_device(const Image& plane, double sigmaColor, double sigmaSpace) {
    if (plane.channels() != 1 || plane.depth() != NLE_8U) throw std::runtime_error("bilateralFilter8: 8UC1 image expected");
    nle_ctx* c = shared_ctx();
    const size_t n = plane.total();
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)plane.ptr<unsigned char>()[i];
    Dev d_in(c, n * 4), d_out(c, n * 4);
    check(nle_dev_upload(c, d_in.p, h.data(), n * 4), c);
    check(nle_bilateral8(c, d_in.f(), plane.rows, plane.cols, sigmaColor, sigmaSpace, d_out.f()), c);
    check(nle_dev_download(c, h.data(), d_out.p, n * 4), c);
    Image out(plane.rows, plane.cols, NLE_8U, 1);
    for (size_t i = 0; i < n; ++i) out.ptr<unsigned char>()[i] = (unsigned char)h[i];
    return out;
}

void NLEFilter::trainForDenoise(const Image& image, int nRowSamples, int nColSamples, DType hx, DType hy,
                                int nSinkhornIter, int nEigenVectors, int sigmaColor, int sigmaSpace) {  // :521-538
    if (image.channels() != 3 || image.depth() != NLE_8U) throw std::runtime_error("Can only enchance RGB image.");
    if (nRowSamples > image.rows || nColSamples > image.cols)
        throw std::runtime_error("Number of samples per row and col must be <= that of image.");
    // BGR -> Lab, L, bilateral filter (8 bit), convertTo(double), trainFilter -- all on the device
    ctx_ = shared_ctx();
    const size_t n = image.total();
    Dev d_bgr(ctx_, n * 3), d_L(ctx_, n * 4), d_Y(ctx_, n * 4);
    check(nle_dev_upload(ctx_, d_bgr.p, image.ptr<unsigned char>(), n * 3), ctx_);
    check(nle_bgr2lab8(ctx_, static_cast<unsigned char*>(d_bgr.p), (long long)n, nullptr, d_L.f()), ctx_);
    check(nle_bilateral8(ctx_, d_L.f(), image.rows, image.cols, sigmaColor, sigmaSpace, d_Y.f()), ctx_);
    trainOnDevice(d_Y.f(), image.rows, image.cols, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors);
}

Image NLEFilter::denoise(const Image& image, DType k, int sigmaColor, int sigmaSpace) const {  // :349-410
    if (image.channels() != 3) throw std::runtime_error("Can only enchance RGB image.");  // (sic) :351-353
    long long n = 0;
    if (f_) nle_filter_info(f_, &n, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (!f_ || (long long)image.total() != n)
        throw std::runtime_error(
            "Cannot apply filter on image with different size from the image filter was trained on.");
    if (image.depth() != NLE_8U) throw std::runtime_error("Can only enchance RGB image.");
    const size_t np = image.total();
    Dev d_bgr(ctx_, np * 3), d_lab(ctx_, np * 3), d_L(ctx_, np * 4), d_Y(ctx_, np * 4), d_c(ctx_, np * 4),
        d_a(ctx_, np * 4), d_b(ctx_, np * 4);
    check(nle_dev_upload(ctx_, d_bgr.p, image.ptr<unsigned char>(), np * 3), ctx_);
    check(nle_bgr2lab8(ctx_, static_cast<unsigned char*>(d_bgr.p), (long long)np, static_cast<unsigned char*>(d_lab.p),
                       d_L.f()), ctx_);
    // the L channel becomes its bilateral-filtered version (:370-373; `apply` on it is commented out, :389)
    check(nle_bilateral8(ctx_, d_L.f(), image.rows, image.cols, sigmaColor, sigmaSpace, d_Y.f()), ctx_);
    Vec t = eigvals();
    for (int i = 0; i < t.size(); ++i) {  // :380-387
        const DType ev = std::min(t(i), 1.0);
        if (verbose) std::cout << "eig " << i << " val: " << ev << std::endl;
        t(i) = std::pow(ev, k);
    }
    for (int ch = 1; ch <= 2; ++ch) {  // :390-391
        float* d_out = ch == 1 ? d_a.f() : d_b.f();
        check(nle_lab8_channel(ctx_, static_cast<unsigned char*>(d_lab.p), (long long)np, ch, d_c.f()), ctx_);
        check(nle_apply_rounded8(f_, d_c.f(), image.rows, image.cols, t.data(), d_out), ctx_);  // + :394-399
    }
    // max(0) / min(255) / convertTo(CV_8U) of the three planes, merge, Lab -> BGR (:393-409)
    check(nle_lab2bgr8_planes(ctx_, static_cast<unsigned char*>(d_lab.p), d_Y.f(), d_a.f(), d_b.f(), (long long)np,
                              static_cast<unsigned char*>(d_bgr.p)), ctx_);
    Image out(image.rows, image.cols, NLE_8U, 3);
    check(nle_dev_download(ctx_, out.ptr<unsigned char>(), d_bgr.p, np * 3), ctx_);
    return out;
}

Image NLEFilter::apply(const Image& channel, const Vec& transformedEigVals) const {  // :445-458
    long long n = 0;
    if (f_) nle_filter_info(f_, &n, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (!f_ || (long long)channel.total() != n)
        throw std::runtime_error("Number of values in channel must match that of training image.");
    int K = 0;
    nle_filter_info(f_, nullptr, &K, nullptr, nullptr, nullptr, nullptr);
    if (transformedEigVals.size() != K) throw std::runtime_error("apply: one transformed eigenvalue per eigenvector expected");
    std::vector<float> x = plane_f32(channel), y(channel.total());
    check(nle_apply_host(f_, x.data(), channel.rows, channel.cols, transformedEigVals.data(), y.data()), ctx_);
    Image out(channel.rows, channel.cols, NLE_64F, 1);
    double* d = out.ptr<double>();
    for (size_t i = 0; i < y.size(); ++i) d[i] = y[i];
    return out;
}

std::vector<Image> NLEFilter::applyLayers(const Image& channel, int nLayers) const {
    long long n = 0;
    if (f_) nle_filter_info(f_, &n, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (!f_ || (long long)channel.total() != n)
        throw std::runtime_error("Number of values in channel must match that of training image.");
    std::vector<float> x = plane_f32(channel), y((size_t)nLayers * channel.total());
    check(nle_apply_layers_host(f_, x.data(), channel.rows, channel.cols, nLayers, y.data()), ctx_);
    std::vector<Image> out;
    for (int l = 0; l < nLayers; ++l) {
        Image m(channel.rows, channel.cols, NLE_64F, 1);
        double* d = m.ptr<double>();
        for (size_t i = 0; i < channel.total(); ++i) d[i] = y[(size_t)l * channel.total() + i];
        out.push_back(std::move(m));
    }
    return out;
}

// ---- NLE_DEVICES: the two calls of the `enhance` CLI over a device group (row slabs, slab input) ----
void NLEFilter::trainForEnhancementGroup(const Image& image, int nRowSamples, int nColSamples, DType hx, DType hy,
                                         int nSinkhornIter, int nEigenVectors) {
    int rs, ro, nr, cs, co, nc;
    check(nle_sample_grid(image.rows, image.cols, nRowSamples, nColSamples, &rs, &ro, &nr, &cs, &co, &nc), nullptr);
    DeviceGroup* g = device_group(nr * nc);
    const int G = g->size(), H = image.rows, W = image.cols;
    if (G > H) throw std::runtime_error("more devices than image rows");
    fh_.reset();
    f_ = nullptr;
    group_.assign(G, nullptr);
    if (verbose) {
        std::cout << "Computing kernel" << std::endl;
        std::cout << "Nystrom approximation" << std::endl;
        std::cout << "Sinkhorn" << std::endl;
        std::cout << "Orthogonalize" << std::endl;
    }
    std::vector<nle_filter*> fs(G, nullptr);
    g->run([&](int r) {
        nle_ctx* c = g->ctx[r];
        int r0 = 0, r1 = 0;
        check(nle_slab_rows(H, r, G, &r0, &r1), c);
        const size_t nl = (size_t)(r1 - r0) * W;
        Dev d_bgr(c, nl * 3), d_L(c, nl * 4);
        check(nle_dev_upload(c, d_bgr.p, image.ptr<unsigned char>() + (size_t)r0 * W * 3, nl * 3), c);
        check(nle_bgr2lab8(c, static_cast<unsigned char*>(d_bgr.p), (long long)nl, nullptr, d_L.f()), c);
        check(nle_train(c, d_L.f(), H, W, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors, &fs[r]), c);
    });
    for (int r = 0; r < G; ++r) group_[r].reset(fs[r], [](nle_filter* f) { nle_filter_destroy(f); });
    ctx_ = g->ctx[0];
    fh_ = group_[0];
    f_ = fh_.get();
    rows_ = H;
    cols_ = W;
    if (verbose) {
        Vec ev = eigvals();
        const int nshow = std::min(std::min(nEigenVectors, 5), ev.size());
        double mn[5], mx[5];
        for (int i = 0; i < nshow; ++i) mn[i] = 1e300, mx[i] = -1e300;
        for (int r = 0; r < G && nshow > 0; ++r) {  // min / max over the ranks' slabs
            double a[5], b[5];
            check(nle_filter_eigvec_range(group_[r].get(), nshow, a, b), g->ctx[r]);
            for (int i = 0; i < nshow; ++i) mn[i] = std::min(mn[i], a[i]), mx[i] = std::max(mx[i], b[i]);
        }
        for (int i = 0; i < nshow; i++)
            std::cout << "Eigvec " << i << " eigval: " << ev(i) << " minCoeff: " << mn[i] << " maxCoeff: " << mx[i]
                      << std::endl;
    }
}

Image NLEFilter::enhanceGroup(const Image& image, const std::vector<DType>& weights) const {
    DeviceGroup* g = device_group(-1);
    const int G = (int)group_.size(), H = image.rows, W = image.cols;
    if (!g || G != g->size() || H != rows_ || W != cols_)
        throw std::runtime_error(
            "Cannot apply filter on image with different size from the image filter was trained on.");
    const Vec fS = transformEigenValues(eigvals(), weights);
    Image out(H, W, NLE_8U, 3);
    g->run([&](int r) {
        nle_ctx* c = g->ctx[r];
        int r0 = 0, r1 = 0;
        check(nle_slab_rows(H, r, G, &r0, &r1), c);
        const size_t nl = (size_t)(r1 - r0) * W;
        Dev d_bgr(c, nl * 3), d_lab(c, nl * 3), d_L(c, nl * 4), d_y(c, nl * 4);
        check(nle_dev_upload(c, d_bgr.p, image.ptr<unsigned char>() + (size_t)r0 * W * 3, nl * 3), c);
        check(nle_bgr2lab8(c, static_cast<unsigned char*>(d_bgr.p), (long long)nl, static_cast<unsigned char*>(d_lab.p),
                           d_L.f()), c);
        check(nle_apply_rounded8(group_[r].get(), d_L.f(), H, W, fS.data(), d_y.f()), c);  // :431-436
        check(nle_lab2bgr8(c, static_cast<unsigned char*>(d_lab.p), d_y.f(), (long long)nl,
                           static_cast<unsigned char*>(d_bgr.p)), c);
        check(nle_dev_download(c, out.ptr<unsigned char>() + (size_t)r0 * W * 3, d_bgr.p, nl * 3), c);
    });
    return out;
}

Image NLEFilter::enhance(const Image& image, const std::vector<DType>& weights) const {  // :412-443
    if (image.channels() != 3) throw std::runtime_error("Can only enhance RGB image.");
    if (!group_.empty()) {
        if (image.depth() != NLE_8U) throw std::runtime_error("Can only enhance RGB image.");
        return enhanceGroup(image, weights);
    }
    long long n = 0;
    if (f_) nle_filter_info(f_, &n, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (!f_ || (long long)image.total() != n)
        throw std::runtime_error(
            "Cannot apply filter on image with different size from the image filter was trained on.");
    if (image.depth() != NLE_8U) throw std::runtime_error("Can only enhance RGB image.");
    // :422-440 on the device: BGR -> Lab, L -> float, apply, clamp + round to 8 bit, merge with a, b, Lab -> BGR
    const size_t np = image.total();
    Dev d_bgr(ctx_, np * 3), d_lab(ctx_, np * 3), d_L(ctx_, np * 4), d_y(ctx_, np * 4);
    check(nle_dev_upload(ctx_, d_bgr.p, image.ptr<unsigned char>(), np * 3), ctx_);
    check(nle_bgr2lab8(ctx_, static_cast<unsigned char*>(d_bgr.p), (long long)np, static_cast<unsigned char*>(d_lab.p),
                       d_L.f()), ctx_);
    Vec fS = transformEigenValues(eigvals(), weights);
    check(nle_apply_rounded8(f_, d_L.f(), image.rows, image.cols, fS.data(), d_y.f()), ctx_);  // :431-436
    check(nle_lab2bgr8(ctx_, static_cast<unsigned char*>(d_lab.p), d_y.f(), (long long)np,
                       static_cast<unsigned char*>(d_bgr.p)), ctx_);
    Image out(image.rows, image.cols, NLE_8U, 3);
    check(nle_dev_download(ctx_, out.ptr<unsigned char>(), d_bgr.p, np * 3), ctx_);
    return out;
}

Vec NLEFilter::eigvals() const {
    if (!f_) return Vec();
    int K = 0;
    nle_filter_info(f_, nullptr, &K, nullptr, nullptr, nullptr, nullptr);
    Vec v(K);
    nle_filter_eigvals(f_, v.data());
    return v;
}

Mat NLEFilter::eigvecs() const {
    if (!f_) return Mat();
    long long n = 0;
    int K = 0;
    nle_filter_info(f_, &n, &K, nullptr, nullptr, nullptr, nullptr);
    const float* d_V = nullptr;
    int ld = 0;
    nle_filter_eigvecs(f_, &d_V, &ld);
    std::vector<float> h((size_t)n * ld);
    check(nle_dev_download(ctx_, h.data(), d_V, h.size() * 4), ctx_);
    Mat V((int)n, K);
    for (int k = 0; k < K; ++k)
        for (long long i = 0; i < n; ++i) V((int)i, k) = h[(size_t)i * ld + k];
    return V;
}

void NLEFilter::timings(double ms[6]) const {
    if (f_) nle_filter_timings(f_, ms);
}

void NLEFilter::diag(int info[8]) const {
    std::fill(info, info + 8, 0);
    if (f_) nle_filter_diag(f_, info);
}

}  // namespace nle
