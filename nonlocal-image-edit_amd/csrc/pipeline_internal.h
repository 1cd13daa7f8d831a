// Internals shared by the host-side translation units of libnle_hip.so (pipeline.hip, devsolve.hip): the ctx / filter
// structs behind the opaque handles of include/nle.h, error plumbing, the stream-ordered workspace arena, per-kernel event
// timing, the sample-grid closed form and the RCCL loader.  Not installed, not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/nle.h"
#include "eigen_sym.h"
#include "kernels.h"

using nlek::GridSpec;

// ------------------------------------------------------------------------------ types
struct nle_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int rank = 0, world = 1;
    nle_allreduce_fn allreduce = nullptr;
    void* ar_user = nullptr;
    ncclComm_t comm = nullptr;  // native RCCL (nle_ctx_init_rccl / nle_ctx_set_rccl_comm): all-reduce in place on `stream`
    bool own_comm = false;
    std::atomic<int> comm_aborted{0};  // nle_ctx_abort_rccl (possibly from another thread): collectives fail from here on
    std::mutex comm_mu;                // serialises the enqueue of a collective on `comm` against nle_ctx_abort_rccl, which
                                       // frees it (ncclCommAbort) on another thread: no enqueue may use a freed communicator
    double* d_comm = nullptr;
    size_t comm_len = 0;
    std::string err;
    // per-kernel HIP-event timing (nle_ctx_profile): records are resolved at the next
    // point where the stream is synchronised anyway
    // workspace cache: device buffers released by a call are kept and handed to the next call
    // (hipMalloc/hipFree of multi-GB buffers cost milliseconds and hipFree synchronises the device);
    // everything is stream-ordered on `stream`, so reuse needs no extra synchronisation
    std::multimap<size_t, void*> arena_free;
    size_t arena_bytes = 0;
    double* d_lut = nullptr;        // sRGB decode table of the colour wrapper
    std::set<nle_filter*> filters;  // live filters trained on this ctx (orphaned if the ctx dies first)
    hipEvent_t aux_ev = nullptr;
    void* h_stage = nullptr;  // page-locked staging block for the solvers' larger transfers (pinned_take())
    size_t h_stage_bytes = 0, h_stage_used = 0, h_stage_want = 0;
    hipStream_t aux_stream = nullptr;   // second compute stream (devsolve.hip: the root of Wa beside the Gram kernels)
    hipStream_t copy_stream = nullptr;  // device-to-host copies of finished output layers (host-buffer entry points)
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    int mode = 0;  // nle_ctx_set_mode: 0 auto, 1 materialised Phi, 2 Phi-free, 3 Phi-free without look-up tables
    bool slab_input = false;  // nle_ctx_set_slab_input: planes handed in hold this rank's rows only
    bool nystrom_bf16x3 = false;  // nle_ctx_set_nystrom_bf16x3: the fused Nystrom GEMM on the bf16 MFMA with split operands
    int topk_solver = 0;  // nle_ctx_set_topk_solver: 0 full eigensolve of Q (:313-316), 1 Lanczos top-K (:170-199)
    bool profiling = false;
    bool profile_all = false;  // level 2: also the small / second-stage kernels (each timed launch costs ~10 us of gaps)
    struct ProfRec {
        int kid;
        hipEvent_t a, b;
    };
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    long long prof_launches[NLE_KERNEL_COUNT] = {0};
    double prof_ms[NLE_KERNEL_COUNT] = {0};
};

struct nle_filter {
    nle_ctx* ctx = nullptr;
    int H = 0, W = 0, row0 = 0, row1 = 0;
    long long n_local = 0;
    int K = 0, ldv = 0, r = 0, p = 0;
    float* d_V = nullptr;  // m_eigvecs (n_local x ldv fp32); in the lazy form it is materialised on first request
    size_t v_bytes = 0;
    double* d_V64 = nullptr;  // fp64 formulation (generic64.hip): m_eigvecs in fp64, same leading dimension
    size_t v64_bytes = 0;
    std::vector<double> eigvals;
    double ms[6] = {0, 0, 0, 0, 0, 0};
    // nle_filter_diag: formulation taken, eigenvalues kept by the three cuts (:214 on Ka, Wa, Q), Cholesky shortcuts
    int formulation = 0, r_wa = 0, r_q = 0, chol_ka = 0, chol_wa = 0;
    // Lazy / sample-space form (tables formulation): m_eigvecs is the implicit V = diag(c) K D.  apply()
    // works on the p-sized side of it (t = D^T sum_i k_i c_i x_i, y = c_i k_i . D(f o t)) and never needs
    // the N x K' matrix; nle_filter_eigvecs & co. build it on demand with the projection kernel.
    bool lazy = false;
    nlek::GridSpec gs{};
    float nsw = 0.f, npw = 0.f;
    int ldd = 0, P64 = 0;
    float* d_plane = nullptr;  // nle_train_host: the uploaded training plane (full image), kept for apply(h_x == NULL)
    size_t plane_bytes = 0;
    float* d_lum = nullptr;  // this rank's slab of the training luminance
    double *d_c = nullptr, *d_er = nullptr, *d_ecT = nullptr, *d_Ep = nullptr, *d_D = nullptr, *d_Vrows = nullptr;
    float4* d_samples = nullptr;
    long long *d_sample_pix = nullptr, *d_sample_loc = nullptr;
    bool has_sorted = false;  // level-sorted rows (sorted.hip) of the training plane, for the apply's reduce half
    nlek::SortedRows sorted{};
    std::vector<std::pair<void*, size_t>> owned;  // workspace-cache buffers that live as long as the filter
    std::vector<double> h_Vrows;                  // p x K col-major: exact rows of V at the sample pixels
    std::vector<long long> h_sample_pix;
};


namespace nlep {

struct Fail {
    int code;
    std::string msg;
};

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw Fail{NLE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)};           \
    } while (0)

// ---- device workspace: served from the ctx's cache when a call is in progress (g_cur) ----
inline thread_local nle_ctx* g_cur = nullptr;
struct CurCtx {
    nle_ctx* prev;
    explicit CurCtx(nle_ctx* c) : prev(g_cur) { g_cur = c; }
    ~CurCtx() { g_cur = prev; }
};

inline void* arena_alloc(nle_ctx* c, size_t bytes) {
    if (c) {
        auto it = c->arena_free.lower_bound(bytes);
        if (it != c->arena_free.end() && it->first <= bytes + bytes / 4 + 4096) {  // close enough fit
            void* p = it->second;
            c->arena_free.erase(it);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess && c && !c->arena_free.empty()) {  // out of memory: drop the cache and retry
        (void)hipStreamSynchronize(c->stream);
        for (auto& kv : c->arena_free) (void)hipFree(kv.second);
        c->arena_free.clear();
        c->arena_bytes = 0;
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) throw Fail{NLE_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)};
    if (c) c->arena_bytes += bytes;
    return p;
}

inline void arena_release(nle_ctx* c, void* p, size_t bytes) {
    if (!p) return;
    if (c) {
        c->arena_free.emplace(bytes, p);
    } else {
        (void)hipFree(p);
    }
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    nle_ctx* owner = nullptr;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void alloc(size_t count) {
        release();
        if (count) {
            owner = g_cur;
            p = static_cast<T*>(arena_alloc(owner, count * sizeof(T)));
        }
        n = count;
    }
    void release() {
        if (p) arena_release(owner, p, n * sizeof(T));
        p = nullptr;
        n = 0;
    }
    T* take() {  // ownership moves to the caller (bytes = n * sizeof(T), release with arena_release)
        T* q = p;
        p = nullptr;
        n = 0;
        return q;
    }
    ~DevBuf() { release(); }
};

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit Timer(hipStream_t st) : s(st) {
        HIP_OK(hipEventCreate(&a));
        HIP_OK(hipEventCreate(&b));
    }
    ~Timer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    void start() { HIP_OK(hipEventRecord(a, s)); }
    void stop() { HIP_OK(hipEventRecord(b, s)); }
    double ms() {
        HIP_OK(hipEventSynchronize(b));
        float t = 0.f;
        HIP_OK(hipEventElapsedTime(&t, a, b));
        return t;
    }
};

inline int ld4(int n) { return (n + 3) & ~3; }

// exp(-d2/hx^2 - dv^2/hy^2) = exp2(nsw*d2 + npw*dv^2)  (reference src/filter.cpp:128-129,144-145)
constexpr double kLog2e = 1.4426950408889634074;
inline float nsw_of(double h) { return (float)(-kLog2e / (h * h)); }

// ---- per-kernel event timing (nle_ctx_profile) ----
inline hipEvent_t prof_event(nle_ctx* c) {
    if (!c->prof_pool.empty()) {
        hipEvent_t e = c->prof_pool.back();
        c->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    HIP_OK(hipEventCreate(&e));
    return e;
}

struct Prof {
    nle_ctx* c;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(nle_ctx* c_, int kid_) : c(c_), kid(kid_) {
        if (!c->profiling) return;
        if (!c->profile_all && (kid == NLE_K_SMALL || kid == NLE_K_REDUCE || kid == NLE_K_SINK_TABLES)) return;
        a = prof_event(c);
        b = prof_event(c);
        HIP_OK(hipEventRecord(a, c->stream));
    }
    void end() {
        if (!a) return;
        HIP_OK(hipEventRecord(b, c->stream));
        c->prof_pending.push_back({kid, a, b});
        a = nullptr;
    }
};

#define PROFILED(ctx_, kid_, expr)  \
    do {                            \
        Prof pf_((ctx_), (kid_));   \
        HIP_OK(expr);               \
        pf_.end();                  \
    } while (0)

// times the kernels of a composite launcher separately (kernels.h: LaunchObserver)
struct ProfObserver : nlek::LaunchObserver {
    nle_ctx* c;
    const int* map;
    Prof* cur = nullptr;
    ProfObserver(nle_ctx* c_, const int* map_) : c(c_), map(map_) {}
    void begin(int sub) override { cur = new Prof(c, map[sub]); }
    void end() override {
        if (cur) {
            cur->end();
            delete cur;
            cur = nullptr;
        }
    }
    ~ProfObserver() override { end(); }
};

// resolve pending records; the caller has synchronised the stream
inline void prof_flush(nle_ctx* c) {
    for (auto& r : c->prof_pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            c->prof_ms[r.kid] += t;
            c->prof_launches[r.kid] += 1;
        }
        c->prof_pool.push_back(r.a);
        c->prof_pool.push_back(r.b);
    }
    c->prof_pending.clear();
}

inline bool make_grid(int H, int W, int nRow, int nCol, GridSpec* gs) {
    // samplePixels, reference src/filter.cpp:56-71, closed form
    if (H <= 0 || W <= 0 || nRow <= 0 || nCol <= 0 || nRow > H || nCol > W) return false;
    gs->H = H;
    gs->W = W;
    gs->rowStep = H / nRow;
    gs->colStep = W / nCol;
    gs->rowOff = (gs->rowStep - 1 + (H - gs->rowStep * nRow)) / 2;
    gs->colOff = (gs->colStep - 1 + (W - gs->colStep * nCol)) / 2;
    // r >= off, r <= H - off, (r - off) % step == 0, r < H
    auto count = [](int n, int off, int step) {
        const int hi = std::min(n - 1, n - off);
        if (hi < off) return 0;
        return (hi - off) / step + 1;
    };
    gs->nSelRows = count(H, gs->rowOff, gs->rowStep);
    gs->nSelCols = count(W, gs->colOff, gs->colStep);
    return gs->nSelRows > 0 && gs->nSelCols > 0;
}

inline void slab(int H, int rank, int world, int* row0, int* row1) {
    *row0 = (int)(((long long)rank * H) / world);
    *row1 = (int)(((long long)(rank + 1) * H) / world);
}

inline double recip0(double v, double eps = NLE_EPS) { return std::fabs(v) >= eps ? 1.0 / v : 0.0; }

inline void check_image_size(int H, int W) {
    if (H <= 0 || W <= 0) throw Fail{NLE_ERR_INVALID, "image must be non-empty"};
    if ((long long)H * W >= (1ll << 31)) throw Fail{NLE_ERR_INVALID, "image too large (H*W must be < 2^31)"};
}

// RCCL is loaded on first use (librccl.so is half a gigabyte: a single-GPU `enhance` never pays for it); if the host
// process already has it (torch.distributed), dlopen by soname returns that same instance
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
inline RcclApi& rccl() {
    // resolved once, thread-safely (function-local static), and published only when every symbol is there: a failed
    // attempt throws out of the initialiser and is retried by the next caller
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) throw Fail{NLE_ERR_COMM, std::string("cannot load librccl.so: ") + dlerror()};
        auto sym = [&](const char* n) {
            void* p = dlsym(a.lib, n);
            if (!p) throw Fail{NLE_ERR_COMM, std::string("librccl.so lacks ") + n};
            return p;
        };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        return a;
    }();
    return api;
}
#define RCCL_OK(expr)                                                                                           \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) throw Fail{NLE_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r_)}; \
    } while (0)

// sum over ranks of n doubles at device pointer d (stream ordered)
inline void all_reduce(nle_ctx* c, double* d, size_t n) {
    if (c->comm_aborted.load(std::memory_order_acquire))
        throw Fail{NLE_ERR_COMM, "the communicator of this ctx was aborted (another rank failed)"};
    if (c->comm) {  // native: one ncclAllReduce in place on the ctx's stream (also for world == 1: same code path)
        // the ENQUEUE (not the collective's execution) under the mutex: an abort from another thread either comes first --
        // seen by the check below -- or waits for the enqueue to return and then ends the pending collective
        std::lock_guard<std::mutex> lk(c->comm_mu);
        if (c->comm_aborted.load(std::memory_order_acquire))
            throw Fail{NLE_ERR_COMM, "the communicator of this ctx was aborted (another rank failed)"};
        RCCL_OK(rccl().AllReduce(d, d, n, ncclDouble, ncclSum, c->comm, c->stream));
        return;
    }
    if (c->world <= 1) return;
    if (!c->allreduce || !c->d_comm || c->comm_len < n)
        throw Fail{NLE_ERR_COMM, "world > 1 but no all-reduce callback / comm buffer too small"};
    HIP_OK(hipMemcpyAsync(c->d_comm, d, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (c->allreduce(c->ar_user, c->d_comm, n) != 0) throw Fail{NLE_ERR_COMM, "all-reduce callback failed"};
    HIP_OK(hipMemcpyAsync(d, c->d_comm, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
}

// A rank-local verdict that the collectives after it depend on (does Phi fit HERE?  which formulation?) is agreed over the
// ranks before anything acts on it: the number of ranks on which `flag` holds.  Without this a rank that refuses (or picks
// another formulation) leaves its peers blocked in their next all-reduce.  One 8-byte all-reduce; nothing when world == 1.
// (The tests inject a dissenting rank through the all-reduce callback: this is the only one-double all-reduce.)
inline int ranks_where(nle_ctx* c, bool flag) {
    if (c->world <= 1 && !c->comm) return flag ? 1 : 0;
    DevBuf<double> d(1);
    double v = flag ? 1.0 : 0.0;
    HIP_OK(hipMemcpyAsync(d.p, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
    all_reduce(c, d.p, 1);
    HIP_OK(hipMemcpyAsync(&v, d.p, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return (int)(v + 0.5);
}

// Page-locked staging memory of the ctx for the solvers' larger transfers, handed out bump-style within one train call
// (large copies from / to pageable memory leave the runtime with milliseconds of clean-up at a later synchronisation).
// pinned_reset at the start of a call; pinned_take returns nullptr when the block is exhausted (the caller then copies
// from pageable memory; the block is sized to the high-water mark at the next reset).
inline void pinned_reset(nle_ctx* c) {
    if (c->h_stage_want > c->h_stage_bytes) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
        const size_t want = c->h_stage_want + c->h_stage_want / 4;
        if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) == hipSuccess) c->h_stage_bytes = want;
        else c->h_stage = nullptr;
    }
    c->h_stage_used = 0;
    c->h_stage_want = 0;
}
inline void* pinned_take(nle_ctx* c, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    c->h_stage_want += bytes;
    if (!c->h_stage || c->h_stage_used + bytes > c->h_stage_bytes) return nullptr;
    void* p = static_cast<char*>(c->h_stage) + c->h_stage_used;
    c->h_stage_used += bytes;
    return p;
}
// host-to-device copy of n doubles through the staging block when it has room
inline void upload_staged(nle_ctx* c, double* d_dst, const double* h_src, size_t n, hipStream_t st) {
    if (void* hp = pinned_take(c, n * sizeof(double))) {
        std::memcpy(hp, h_src, n * sizeof(double));
        h_src = static_cast<const double*>(hp);
    }
    HIP_OK(hipMemcpyAsync(d_dst, h_src, n * sizeof(double), hipMemcpyHostToDevice, st));
}

inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
// NLE_TRACE=1: stage marks of a train call on stderr
struct Trace {
    bool on;
    double t0, last;
    Trace() : on(std::getenv("NLE_TRACE") != nullptr) { t0 = last = now_ms(); }
    void mark(const char* what) {
        if (!on) return;
        const double t = now_ms();
        std::fprintf(stderr, "[nle trace] %-28s +%8.3f ms  (t=%8.3f)\n", what, t - last, t - t0);
        last = t;
    }
};

// error text of the last failed call without a ctx (nle_ctx_create)
inline thread_local std::string g_create_err;

// body of every C ABI entry point: exceptions become status codes, the ctx's workspace cache serves the DevBufs
template <typename Fn>
int guard(nle_ctx* c, Fn&& fn) {
    CurCtx scope(c);
    try {
        fn();
        return NLE_OK;
    } catch (const Fail& e) {
        if (c) c->err = e.msg; else g_create_err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        if (c) c->err = e.what(); else g_create_err = e.what();
        return NLE_ERR_INVALID;
    }
}

}  // namespace nlep
