// C ABI, the part that holds no filter arithmetic of its own: contexts and their resources, transfers, the settings, the
// RCCL bootstrap, colour conversion and the bilateral pre-filter launches, the host-side eigen-solver entry points, the
// closed-form helpers (sample grid, row slabs, eigenvalue transforms) and the profiling switches.  The train / apply /
// stage-level entry points are in pipeline.hip, the device dense solvers' in devsolve.hip.
#include <mutex>

#include "lab8_fixed.h"
#include "pipeline_internal.h"

using nlek::GridSpec;
using namespace nlep;

extern "C" {

int nle_ld(int n) { return ld4(n); }

size_t nle_comm_len(int n_samples) {
    const int ld = ld4(n_samples);
    return std::max(std::max((size_t)nlek::gram_num_tiles(ld) * 1024, (size_t)nlek::gram64_num_tiles(n_samples) * 256),
                    (size_t)n_samples * n_samples) +
           8 * (size_t)nlek::sink_pass_ld(n_samples);
}

int nle_ctx_create(int device, void* stream, nle_ctx** out) {
    if (!out) return NLE_ERR_INVALID;
    *out = nullptr;
    return guard(nullptr, [&] {
        int ndev = 0;
        HIP_OK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw Fail{NLE_ERR_HIP, "no HIP device (this library has no CPU fallback)"};
        if (device < 0 || device >= ndev) throw Fail{NLE_ERR_INVALID, "device index out of range"};
        HIP_OK(hipSetDevice(device));
        auto c = new nle_ctx();
        c->device = device;
        if (const char* e = std::getenv("NLE_Q_SOLVER")) c->topk_solver = (std::string(e) == "lanczos") ? 1 : 0;
        if (stream) {
            c->stream = reinterpret_cast<hipStream_t>(stream);
        } else {
            hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
            if (e != hipSuccess) {
                delete c;
                throw Fail{NLE_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)};
            }
            c->own_stream = true;
        }
        *out = c;
    });
}

void nle_ctx_destroy(nle_ctx* ctx) {
    if (!ctx) return;
    for (auto& r : ctx->prof_pending) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && ctx->own_comm && !ctx->comm_aborted.load()) (void)rccl().CommDestroy(ctx->comm);
    for (auto* f : ctx->filters) f->ctx = nullptr;  // their V is freed directly when they are destroyed
    if (ctx->d_lut) (void)hipFree(ctx->d_lut);
    for (auto e : ctx->copy_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
        if (ctx->aux_ev) (void)hipEventDestroy(ctx->aux_ev);
    }
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (auto& kv : ctx->arena_free) (void)hipFree(kv.second);
    ctx->arena_free.clear();
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* nle_last_error(const nle_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int nle_ctx_synchronize(nle_ctx* ctx) {
    if (!ctx) return NLE_ERR_INVALID;
    return guard(ctx, [&] { HIP_OK(hipStreamSynchronize(ctx->stream)); });
}

int nle_dev_alloc(nle_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return NLE_ERR_INVALID;
    *d_ptr = nullptr;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(hipMalloc(d_ptr, bytes ? bytes : 1));
    });
}

void nle_dev_free(nle_ctx* ctx, void* d_ptr) {
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ptr);
}

int nle_host_alloc(nle_ctx* ctx, size_t bytes, void** h_ptr) {
    if (!ctx || !h_ptr) return NLE_ERR_INVALID;
    *h_ptr = nullptr;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    });
}

void nle_host_free(nle_ctx* ctx, void* h_ptr) {
    if (!ctx || !h_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipHostFree(h_ptr);
}

int nle_dev_upload(nle_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_dev_download(nle_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!h_dst || !d_src))) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

namespace {
// the tables of the fixed-point 8-bit BGR <-> Lab conversions (lab8_fixed.h), uploaded once per ctx as one blob
const double* colour_lut(nle_ctx* c) {
    if (!c->d_lut) {
        using namespace nlelab8;
        alignas(8) static unsigned char blob[kBlobBytes];
        static std::once_flag once;
        std::call_once(once, [] {
            forward_tables(reinterpret_cast<unsigned short*>(blob + kOffGamma), reinterpret_cast<unsigned short*>(blob + kOffCbrt),
                           reinterpret_cast<int*>(blob + kOffCoeffs));
            inverse_tables(reinterpret_cast<unsigned short*>(blob + kOffYf), reinterpret_cast<unsigned short*>(blob + kOffInvGamma),
                           reinterpret_cast<int*>(blob + kOffInvCoeffs));
        });
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&c->d_lut), kBlobBytes));
        HIP_OK(hipMemcpy(c->d_lut, blob, kBlobBytes, hipMemcpyHostToDevice));
    }
    return c->d_lut;
}
}  // namespace

int nle_lab8_tables(unsigned short* h_gamma, unsigned short* h_cbrt, int* h_coeffs) {
    if (!h_gamma || !h_cbrt || !h_coeffs) return NLE_ERR_INVALID;
    nlelab8::forward_tables(h_gamma, h_cbrt, h_coeffs);
    return NLE_OK;
}

int nle_lab8_inverse_tables(unsigned short* h_yf, int* h_ab_to_xz, unsigned short* h_inv_gamma, int* h_coeffs) {
    if (!h_yf || !h_ab_to_xz || !h_inv_gamma || !h_coeffs) return NLE_ERR_INVALID;
    nlelab8::inverse_tables(h_yf, h_inv_gamma, h_coeffs);
    for (int i = 0; i < nlelab8::kAbN; ++i) h_ab_to_xz[i] = nlelab8::ab_to_xz(i + nlelab8::kMinAB);
    return NLE_OK;
}

int nle_bgr2lab8(nle_ctx* ctx, const unsigned char* d_bgr, long long n, unsigned char* d_lab, float* d_L) {
    if (!ctx || !d_bgr || n < 0 || (!d_lab && !d_L)) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(nlek::bgr2lab8(ctx->stream, d_bgr, n, colour_lut(ctx), d_lab, d_L));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_lab2bgr8(nle_ctx* ctx, const unsigned char* d_lab, const float* d_L, long long n, unsigned char* d_bgr) {
    return nle_lab2bgr8_planes(ctx, d_lab, d_L, nullptr, nullptr, n, d_bgr);
}

int nle_lab2bgr8_planes(nle_ctx* ctx, const unsigned char* d_lab, const float* d_L, const float* d_a, const float* d_b,
                        long long n, unsigned char* d_bgr) {
    if (!ctx || !d_lab || !d_bgr || n < 0) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(nlek::lab2bgr8(ctx->stream, d_lab, d_L, d_a, d_b, n, colour_lut(ctx), d_bgr));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_lab8_channel(nle_ctx* ctx, const unsigned char* d_lab, long long n, int channel, float* d_out) {
    if (!ctx || !d_lab || !d_out || n < 0 || channel < 0 || channel > 2) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(nlek::channel8(ctx->stream, d_lab, n, channel, d_out));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_bilateral_tables(double sigma_color, double sigma_space, int* radius, float* h_space_w, float* h_colour_w) {
    // cv::bilateralFilter with d <= 0 (the reference passes -1, src/filter.cpp:366,371,535)
    if (!radius) return NLE_ERR_INVALID;
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double cc = -0.5 / (sigma_color * sigma_color), sc = -0.5 / (sigma_space * sigma_space);
    const int r = std::max((int)std::lrint(sigma_space * 1.5), 1);
    *radius = r;
    if (!h_space_w && !h_colour_w) return NLE_OK;
    if (!h_space_w || !h_colour_w) return NLE_ERR_INVALID;
    for (int i = 0; i < 256; ++i) h_colour_w[i] = (float)std::exp((double)i * i * cc);
    const int d = 2 * r + 1;
    for (int i = -r; i <= r; ++i)
        for (int j = -r; j <= r; ++j) {
            const double rr = std::sqrt((double)i * i + (double)j * j);
            h_space_w[(i + r) * d + (j + r)] = rr > r ? 0.f : (float)std::exp(rr * rr * sc);
        }
    return NLE_OK;
}

int nle_bilateral8(nle_ctx* ctx, const float* d_src, int H, int W, double sigma_color, double sigma_space, float* d_dst) {
    if (!ctx || !d_src || !d_dst || H < 1 || W < 1 || d_src == d_dst) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        int r = 0;
        nle_bilateral_tables(sigma_color, sigma_space, &r, nullptr, nullptr);
        if (r > nlek::bilateral8_max_radius())
            throw Fail{NLE_ERR_INVALID, "bilateral filter: sigma_space above 42 (radius > 64) is not supported"};
        const int d = 2 * r + 1;
        std::vector<float> sw((size_t)d * d), cw(256);
        nle_bilateral_tables(sigma_color, sigma_space, &r, sw.data(), cw.data());
        // the kernel indexes its 256-entry colour table with |v - v0|: the plane must hold integers 0..255 (CV_8UC1)
        DevBuf<int> d_flag(2);  // verdict, level tiles (unused here)
        int flag = 1;
        HIP_OK(nlek::check_levels(ctx->stream, d_src, (long long)H * W, d_flag.p));
        HIP_OK(hipMemcpyAsync(&flag, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (flag != 0) throw Fail{NLE_ERR_INVALID, "bilateral filter: the plane must be integer valued in [0, 255] (CV_8UC1)"};
        DevBuf<float> d_sw(sw.size()), d_cw(cw.size());
        HIP_OK(hipMemcpyAsync(d_sw.p, sw.data(), sw.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemcpyAsync(d_cw.p, cw.data(), cw.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(nlek::bilateral8(ctx->stream, d_src, H, W, r, d_sw.p, d_cw.p, d_dst));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    });
}

int nle_ctx_trim(nle_ctx* ctx) {
    if (!ctx) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipSetDevice(ctx->device));
        HIP_OK(hipStreamSynchronize(ctx->stream));
        for (auto& kv : ctx->arena_free) (void)hipFree(kv.second);
        ctx->arena_free.clear();
        ctx->arena_bytes = 0;
    });
}

int nle_eigen_decomposition_top(const double* h_M, int n, double eps, int kmax, double* h_U, double* h_D, int* r) {
    if (!h_M || !h_U || !h_D || !r || n < 1 || kmax < 1) return NLE_ERR_INVALID;
    return nleh::eigen_decomposition_top(h_M, n, eps, kmax, h_U, h_D, r) ? NLE_OK : NLE_ERR_NUMERIC;
}

int nle_eigen_decomposition_topk(const double* h_M, int n, double eps, int kmax, double* h_U, double* h_Dk, int* r) {
    if (!h_M || !h_U || !h_Dk || !r || n < 1 || kmax < 1) return NLE_ERR_INVALID;
    return nleh::eigen_decomposition_topk(h_M, n, eps, kmax, h_U, h_Dk, r) ? NLE_OK : NLE_ERR_NUMERIC;
}

int nle_eigen_decomposition_top_device(nle_ctx* ctx, const double* h_M, int n, double eps, int kmax, double* h_U, double* h_D,
                                       int* r) {
    if (!ctx || !h_M || !h_U || !h_D || !r || n < 2 || kmax < 1) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        if (n > nlek::tridiag_max_n()) throw Fail{NLE_ERR_INVALID, "nle_eigen_decomposition_top_device: n exceeds 224"};
        HIP_OK(hipSetDevice(ctx->device));
        const size_t nn = (size_t)n * n;
        DevBuf<double> d_M(nn), d_V(nn), d_t((size_t)3 * n);
        HIP_OK(hipMemcpyAsync(d_M.p, h_M, nn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(nlek::tridiag(ctx->stream, n, d_M.p, nullptr, d_V.p, d_t.p, d_t.p + n, d_t.p + 2 * n));
        std::vector<double> V(nn), t((size_t)3 * n);
        HIP_OK(hipMemcpyAsync(V.data(), d_V.p, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipMemcpyAsync(t.data(), d_t.p, t.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (!nleh::eigen_decomposition_top_reduced(n, eps, std::min(kmax, n), V.data(), t.data(), t.data() + n, t.data() + 2 * n,
                                                   h_U, h_D, r))
            throw Fail{NLE_ERR_NUMERIC, "eigensolver did not converge"};
    });
}

int nle_topk_eigen_decomposition(const double* h_M, int n, int n_largest, double eps, double* h_U, double* h_D, int* r) {
    if (!h_M || !h_U || !h_D || !r || n < 2 || n_largest < 1) return NLE_ERR_INVALID;
    const int nev = std::min(n_largest, n - 1);  // :172
    const int nconv = nleh::lanczos_topk(h_M, n, nev, NLE_EPS, 1000, h_U, h_D, nullptr);
    if (nconv < 0) return NLE_ERR_NUMERIC;
    int k = 0;
    while (k < nconv && h_D[k] >= eps) ++k;  // :186-196
    *r = k;
    return NLE_OK;
}

int nle_ctx_set_slab_input(nle_ctx* ctx, int on) {
    if (!ctx) return NLE_ERR_INVALID;
    ctx->slab_input = on != 0;
    return NLE_OK;
}

int nle_ctx_set_topk_solver(nle_ctx* ctx, int solver) {
    if (!ctx || solver < 0 || solver > 1) return NLE_ERR_INVALID;
    ctx->topk_solver = solver;
    return NLE_OK;
}

int nle_ctx_set_nystrom_bf16x3(nle_ctx* ctx, int on) {
    if (!ctx) return NLE_ERR_INVALID;
    ctx->nystrom_bf16x3 = on != 0;
    return NLE_OK;
}

int nle_ctx_set_mode(nle_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > NLE_MODE_STREAMED_F64) return NLE_ERR_INVALID;
    ctx->mode = mode;
    return NLE_OK;
}

int nle_rccl_unique_id(void* h_id, size_t size) {
    if (!h_id || size < NCCL_UNIQUE_ID_BYTES) return NLE_ERR_INVALID;
    return guard(nullptr, [&] {
        ncclUniqueId id;
        RCCL_OK(rccl().GetUniqueId(&id));
        std::memcpy(h_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    });
}

int nle_ctx_init_rccl(nle_ctx* ctx, int rank, int world, const void* h_id, size_t size) {
    if (!ctx || !h_id || size < NCCL_UNIQUE_ID_BYTES) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        if (world < 1 || rank < 0 || rank >= world) throw Fail{NLE_ERR_INVALID, "bad rank/world"};
        HIP_OK(hipSetDevice(ctx->device));
        if (ctx->comm && ctx->own_comm && !ctx->comm_aborted.load()) (void)rccl().CommDestroy(ctx->comm);
        ctx->comm = nullptr;  // (an aborted communicator has been freed by ncclCommAbort)
        ctx->comm_aborted.store(0);
        ncclUniqueId id;
        std::memcpy(id.internal, h_id, NCCL_UNIQUE_ID_BYTES);
        ncclComm_t comm = nullptr;
        RCCL_OK(rccl().CommInitRank(&comm, world, id, rank));
        ctx->comm = comm;
        ctx->own_comm = true;
        ctx->rank = rank;
        ctx->world = world;
        ctx->allreduce = nullptr;
    });
}

int nle_ctx_abort_rccl(nle_ctx* ctx) {
    if (!ctx) return NLE_ERR_INVALID;
    // no guard(): this may run on another thread than the one that uses the ctx (whose error string it must not touch)
    std::lock_guard<std::mutex> lk(ctx->comm_mu);  // not while the ctx's own thread is inside an enqueue on this communicator
    if (!ctx->comm || !ctx->own_comm) return NLE_OK;
    if (ctx->comm_aborted.exchange(1, std::memory_order_acq_rel) != 0) return NLE_OK;
    try {
        return rccl().CommAbort(ctx->comm) == ncclSuccess ? NLE_OK : NLE_ERR_COMM;
    } catch (...) {
        return NLE_ERR_COMM;
    }
}

int nle_ctx_set_rccl_comm(nle_ctx* ctx, int rank, int world, void* comm) {
    if (!ctx || !comm) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        if (world < 1 || rank < 0 || rank >= world) throw Fail{NLE_ERR_INVALID, "bad rank/world"};
        (void)rccl();  // the all-reduce goes through the loaded library
        if (ctx->comm && ctx->own_comm && !ctx->comm_aborted.load()) (void)rccl().CommDestroy(ctx->comm);
        ctx->comm = reinterpret_cast<ncclComm_t>(comm);
        ctx->comm_aborted.store(0);
        ctx->own_comm = false;
        ctx->rank = rank;
        ctx->world = world;
        ctx->allreduce = nullptr;
    });
}

int nle_ctx_set_shard(nle_ctx* ctx, int rank, int world, nle_allreduce_fn allreduce, void* user,
                      double* d_comm, size_t comm_len) {
    if (!ctx) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        if (world < 1 || rank < 0 || rank >= world) throw Fail{NLE_ERR_INVALID, "bad rank/world"};
        if (world > 1 && (!allreduce || !d_comm || comm_len == 0))
            throw Fail{NLE_ERR_INVALID, "world > 1 needs an all-reduce callback and a comm buffer"};
        ctx->rank = rank;
        ctx->world = world;
        ctx->allreduce = allreduce;
        ctx->ar_user = user;
        ctx->d_comm = d_comm;
        ctx->comm_len = comm_len;
    });
}

int nle_sample_grid(int H, int W, int n_row_samples, int n_col_samples, int* row_step, int* row_off,
                    int* n_sel_rows, int* col_step, int* col_off, int* n_sel_cols) {
    GridSpec gs;
    if (!make_grid(H, W, n_row_samples, n_col_samples, &gs)) return NLE_ERR_INVALID;
    if (row_step) *row_step = gs.rowStep;
    if (row_off) *row_off = gs.rowOff;
    if (n_sel_rows) *n_sel_rows = gs.nSelRows;
    if (col_step) *col_step = gs.colStep;
    if (col_off) *col_off = gs.colOff;
    if (n_sel_cols) *n_sel_cols = gs.nSelCols;
    return NLE_OK;
}

int nle_slab_rows(int H, int rank, int world, int* row0, int* row1) {
    if (H <= 0 || world < 1 || rank < 0 || rank >= world || !row0 || !row1) return NLE_ERR_INVALID;
    slab(H, rank, world, row0, row1);
    return NLE_OK;
}

int nle_eigen_decomposition(const double* h_M, int n, double eps, double* h_U, double* h_D, int* r) {
    if (!h_M || n <= 0 || !h_U || !h_D || !r) return NLE_ERR_INVALID;
    return nleh::eigen_decomposition(h_M, n, eps, h_U, h_D, r) ? NLE_OK : NLE_ERR_NUMERIC;
}

int nle_transform_eigenvalues(const double* h_eigvals, int K, const double* h_weights, int L, double* h_fS) {
    if (!h_eigvals || !h_weights || !h_fS || K < 0 || L < 1) return NLE_ERR_INVALID;
    for (int i = 0; i < K; ++i) {  // reference src/filter.cpp:338-344
        double v = h_weights[0];
        for (int k = 1; k < L; ++k) v += (h_weights[k] - h_weights[k - 1]) * std::pow(h_eigvals[i], (double)k);
        h_fS[i] = v;
    }
    return NLE_OK;
}

static const char* const kKernelNames[NLE_KERNEL_COUNT] = {
    "affinity", "nystrom_extend", "sinkhorn_pass", "reduce_partials", "gram",
    "project",  "apply_reduce",   "apply_expand",  "small",           "sink_tables", "gram_rows",
    "gram_gemm"};

const char* nle_kernel_name(int kid) { return (kid >= 0 && kid < NLE_KERNEL_COUNT) ? kKernelNames[kid] : ""; }

int nle_ctx_profile(nle_ctx* ctx, int enable) {
    if (!ctx) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipStreamSynchronize(ctx->stream));
        prof_flush(ctx);
        ctx->profiling = enable != 0;
        ctx->profile_all = enable >= 2;
        for (int k = 0; k < NLE_KERNEL_COUNT; ++k) {
            ctx->prof_launches[k] = 0;
            ctx->prof_ms[k] = 0.0;
        }
    });
}

int nle_ctx_kernel_stats(nle_ctx* ctx, int kid, long long* launches, double* total_ms) {
    if (!ctx || kid < 0 || kid >= NLE_KERNEL_COUNT) return NLE_ERR_INVALID;
    return guard(ctx, [&] {
        HIP_OK(hipStreamSynchronize(ctx->stream));
        prof_flush(ctx);
        if (launches) *launches = ctx->prof_launches[kid];
        if (total_ms) *total_ms = ctx->prof_ms[kid];
    });
}

}  // extern "C"
