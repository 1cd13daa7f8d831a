// Host side of the device dense solvers (dense64.hip): the symmetric eigen-computation split as
//   device  Householder reduction (one persistent launch) + all eigenvalues by Sturm bisection
//   host    inverse iteration on the tridiagonal matrix for the few eigenvectors wanted (eigen_sym.cpp, O(n k))
//   device  back-transformation of those vectors
// and the Cholesky factor with its inverse.  Used by the train pipeline for the p x p problems of `eigenDecomposition`
// (reference src/filter.cpp:204-228 at :287 and :313) and the Cholesky shortcuts when p is large (devsolve.h says when),
// and exported for the parity tests (nle_sym_eigen_device, nle_cholesky_device).
#include "devsolve.h"

#include <condition_variable>
#include <mutex>

namespace nlep {

namespace {
// Workgroups of persistent launches must all be resident at once: at most (compute units of the device - 32) of them are
// in flight per DEVICE (each needs a compute unit of its own; ctxs on other host threads wait here, not on the GPU)
constexpr int kMaxDevices = 64;
std::mutex g_cu_mu;
std::condition_variable g_cu_cv;
int g_cu_used[kMaxDevices] = {};
}  // namespace

int device_cu_count(int device) {
    static std::mutex mu;
    static int cache[kMaxDevices] = {};
    const int slot = device >= 0 && device < kMaxDevices ? device : 0;
    std::lock_guard<std::mutex> lk(mu);
    if (cache[slot] == 0) {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 1;
        cache[slot] = ncu;
    }
    return cache[slot];
}

CuLease::CuLease(int device, int n_) : dev(device >= 0 && device < kMaxDevices ? device : 0) {
    const int budget = std::max(16, device_cu_count(device) - 32);
    n = std::min(n_, budget);
    std::unique_lock<std::mutex> lk(g_cu_mu);
    g_cu_cv.wait(lk, [&] { return g_cu_used[dev] + n <= budget; });
    g_cu_used[dev] += n;
}
CuLease::~CuLease() {
    {
        std::lock_guard<std::mutex> lk(g_cu_mu);
        g_cu_used[dev] -= n;
    }
    g_cu_cv.notify_all();
}

int dev_solver_min_n() {
    static const int v = [] {
        const char* e = std::getenv("NLE_DEV_SOLVER_MIN");
        return e ? std::max(3, std::atoi(e)) : 288;
    }();
    return v;
}
bool use_dev_solver(int n) {
    return n >= dev_solver_min_n() && n <= nlek::sytrd_max_n() && std::getenv("NLE_HOST_SOLVER") == nullptr;
}

hipStream_t aux_stream(nle_ctx* c) {
    if (!c->aux_stream) {
        HIP_OK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&c->aux_ev, hipEventDisableTiming));
    }
    return c->aux_stream;
}

void DevSymEig::prepare(nle_ctx* c, int n_, hipStream_t stream) {
    if (n_ < 3 || n_ > nlek::sytrd_max_n()) throw Fail{NLE_ERR_INVALID, "device eigensolver: order out of range"};
    st = stream ? stream : c->stream;
    if (n == n_ && pub.p) return;
    n = n_;
    pub.alloc(nlek::sytrd_pub_elems(n));
    tde.alloc((size_t)3 * n);
    status.alloc(1);
}

bool DevSymEig::reduce(nle_ctx* c, int n_, const double* d_M, const double* d_diag_add) {
    if (!pub.p || n != n_) prepare(c, n_, st);
    int G = nlek::sytrd_groups(n);
    if (const char* e = std::getenv("NLE_SYTRD_G")) G = std::atoi(e);
    // every workgroup of the persistent launch needs a compute unit of its own for the whole launch: a device that exposes
    // fewer (a partitioned or smaller part) cannot run it -- the caller takes the host solver
    if (G <= 0 || G + 16 > device_cu_count(c->device)) return false;
    d.assign(n, 0.0);
    e.assign(n, 0.0);
    D.assign(n, 0.0);
    int h_status = 0;
    {
        CuLease lease(c->device, G);
        HIP_OK(nlek::sytrd_dist(st, n, G, d_M, d_diag_add, pub.p, tde.p, tde.p + n, status.p));
        HIP_OK(nlek::tridiag_bisect(st, n, tde.p, tde.p + n, tde.p + 2 * n));
        std::vector<double> hv;
        double* h = static_cast<double*>(pinned_take(c, (size_t)3 * n * sizeof(double)));
        if (!h) {
            hv.resize((size_t)3 * n);
            h = hv.data();
        }
        HIP_OK(hipMemcpyAsync(h, tde.p, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(&h_status, status.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        std::copy(h, h + n, d.begin());
        std::copy(h + n, h + 2 * n, e.begin());
        std::copy(h + 2 * n, h + 3 * n, D.begin());
    }
    // a hand-off between workgroups timed out (the workgroups were not co-resident after all), or the matrix was not
    // finite: not an error of the train -- the host solver (nleh::eigen_decomposition*) takes over
    if (h_status != 0) {
        if (std::getenv("NLE_TRACE")) std::fprintf(stderr, "[nle trace] device eigensolver (n = %d): hand-off timed out, host solver takes over\n", n);
        return false;
    }
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(D[i])) return false;
    return true;
}

void DevSymEig::vectors(nle_ctx* c, int first, int count, double* d_Z) {
    if (count <= 0) return;
    if (first < 0 || first + count > n) throw Fail{NLE_ERR_INVALID, "device eigensolver: eigenvector range"};
    static const bool trace = std::getenv("NLE_EIG_TRACE") != nullptr;
    const double t0 = trace ? now_ms() : 0.0;
    double* hz = static_cast<double*>(pinned_take(c, (size_t)n * count * sizeof(double)));
    if (!hz) {
        hZ.resize((size_t)n * count);
        hz = hZ.data();
    }
    if (!nleh::tridiag_eigenvectors(n, d.data(), e.data(), D.data(), first, count, hz))
        throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge (tridiagonal eigenvectors)"};
    if (trace) std::fprintf(stderr, "[nle eig] device path n = %d: %d vectors by inverse iteration %.3f ms\n", n, count, now_ms() - t0);
    HIP_OK(hipMemcpyAsync(d_Z, hz, (size_t)n * count * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(nlek::sytrd_back(st, n, pub.p, count, d_Z, n));
}

void DevChol::prepare(nle_ctx* c, int n_, hipStream_t stream) {
    st = stream ? stream : c->stream;
    if (n == n_ && L.p) return;
    n = n_;
    const size_t nn = (size_t)n * n;
    L.alloc(nn);
    Linv.alloc(nn);
    tmp.alloc(nlek::potrf_tmp_elems(n) + 1);
    status.alloc(1);
}

void DevChol::factor(nle_ctx* c, int n_, const double* d_A) {
    if (!L.p || n != n_) prepare(c, n_, st);
    HIP_OK(hipMemsetAsync(status.p, 0, sizeof(int), st));
    HIP_OK(nlek::potrf_inverse(st, n, d_A, L.p, Linv.p, tmp.p, tmp.p + nlek::potrf_tmp_elems(n), status.p));
}

bool DevChol::finish(nle_ctx* c) {
    int h_status = 0;
    double tr = 0.0;
    HIP_OK(hipMemcpyAsync(&h_status, status.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&tr, tmp.p + nlek::potrf_tmp_elems(n), sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    inv_trace = tr;
    ok = h_status == 0 && std::isfinite(tr);
    return ok;
}

}  // namespace nlep

using namespace nlep;

extern "C" {

int nle_sym_eigen_device(nle_ctx* ctx, const double* h_M, int n, double eps, int first, int count, double* h_U, double* h_D,
                         int* r) {
    if (!ctx || !h_M || !h_D || !r || n < 3 || first < 0 || count < 0 || first + count > n || (count > 0 && !h_U))
        return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        if (n > nlek::sytrd_max_n()) throw Fail{NLE_ERR_INVALID, "nle_sym_eigen_device: n exceeds 1152"};
        HIP_OK(hipSetDevice(ctx->device));
        const size_t nn = (size_t)n * n;
        DevBuf<double> d_M(nn), d_Z((size_t)n * std::max(count, 1));
        HIP_OK(hipMemcpyAsync(d_M.p, h_M, nn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        DevSymEig es;
        if (!es.reduce(ctx, n, d_M.p, nullptr))
            throw Fail{NLE_ERR_NUMERIC, "device eigensolver: not enough compute units for the persistent reduction, a hand-off "
                                        "between its workgroups timed out, or a non-finite matrix"};
        std::copy(es.D.begin(), es.D.end(), h_D);
        int k = 0;
        while (k < n && es.D[k] >= eps) ++k;  // src/filter.cpp:213-216
        *r = k;
        if (count > 0) {
            es.vectors(ctx, first, count, d_Z.p);
            HIP_OK(hipMemcpyAsync(h_U, d_Z.p, (size_t)n * count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(hipStreamSynchronize(ctx->stream));
        }
    });
}

int nle_cholesky_device(nle_ctx* ctx, const double* h_M, int n, double* h_L, double* h_Linv, double* inv_trace, int* ok) {
    if (!ctx || !h_M || !h_L || !h_Linv || !inv_trace || !ok || n < 1) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        if (n > 4096) throw Fail{NLE_ERR_INVALID, "nle_cholesky_device: n exceeds 4096"};
        HIP_OK(hipSetDevice(ctx->device));
        const size_t nn = (size_t)n * n;
        DevBuf<double> d_M(nn);
        HIP_OK(hipMemcpyAsync(d_M.p, h_M, nn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        DevChol ch;
        ch.factor(ctx, n, d_M.p);
        *ok = ch.finish(ctx) ? 1 : 0;
        *inv_trace = ch.inv_trace;
        HIP_OK(hipMemcpyAsync(h_L, ch.L.p, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipMemcpyAsync(h_Linv, ch.Linv.p, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

}  // extern "C"
