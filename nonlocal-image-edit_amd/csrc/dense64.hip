// Dense fp64 algebra of order n <= 1152 on the device: what is left of `eigenDecomposition` (reference src/filter.cpp:204-228,
// called at :262, :287, :313) and of the Cholesky shortcuts once everything N-sized is fast -- at p = 400..900 samples the two
// symmetric eigen-computations and the factorisation of K_A were half of a train step on one host core.
//
//   sytrd_dist        Householder tridiagonalisation, ONE persistent launch of G workgroups, every WAVE autonomous: whole
//                     columns dealt cyclically to waves (both triangles stored), vectors in registers, ONE hand-off per
//                     step through global memory, "the data is the flag" (MI355X_MICROARCH.md, visibility: 8-byte
//                     agent-scope atomic stores and loads, every word written once per launch into a buffer pre-filled with
//                     an impossible bit pattern, consumers re-read until no word is unset): the y values of the step and
//                     the next column as it stands; v, tau, beta and w are then formed redundantly by every wave
//                     (k_sytrd_wave below; round 3's form had two hand-offs and five barriers per step).
//   tridiag_bisect    all eigenvalues of the tridiagonal matrix by Sturm counts, one wave per eigenvalue, 64 section points
//                     per round (the interval shrinks 65x per pass over d, e): ~10 rounds.
//   sytrd_back        back-transformation of K eigenvectors of T (one wave per vector, the vector in registers).
//   potrf / trtri     blocked Cholesky factor and its inverse: 32-column panels (one kernel: every workgroup re-factors the
//                     32 x 32 diagonal block in LDS and solves its own rows of the panel), trailing update and the inverse's
//                     block rows on k_gemm64s (fp64 MFMA).
// The eigenvectors of T for the wanted eigenvalues come from the host's inverse iteration (eigen_sym.cpp, O(n k)).
#include "kernels.h"
#include "handoff.h"

#include <algorithm>
#include <utility>
#include <cstdlib>

namespace nlek {

namespace {

using namespace handoff;
constexpr int kSyT = 256;      // threads per workgroup of the reduction
constexpr int kSyRows = 5;     // rows per thread: n <= 1152 < 5 * 256

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// The same sum on the DPP network (row shifts inside the 16-lane rows, then the two row broadcasts): ~20 vector instructions
// instead of six dependent LDS permutes per half of the double (~900 cycles measured in the reduction's step).  Every lane
// receives the total (read from lane 63 through a scalar register).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double x) {
    x = dpp_add<0x111, 0xf>(x);  // row_shr:1
    x = dpp_add<0x112, 0xf>(x);  // row_shr:2
    x = dpp_add<0x114, 0xf>(x);  // row_shr:4
    x = dpp_add<0x118, 0xf>(x);  // row_shr:8   -> lane 15 of every row holds the row's sum
    x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double x, int src_lane /* wave-uniform */) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ tridiagonalisation
// pub: (n - 2) records of 2 n + 2 doubles, one per step k: v_k (entries k+1 .. n-1, v_k[k+1] = 1), tau_k at [n], beta_k at
// [n + 1], then y (entries k+1 .. n-1) from [n + 2].  H_k = I - tau_k v_k v_k^T; T = Q^T A Q, Q = H_0 H_1 .. H_{n-3};
// d_out[i] = T(i, i), e_out[i] = T(i, i-1) (e_out[0] = 0): the layout eigen_sym.cpp's tridiagonal routines take.

// ------------------------------------------------------------------------- tridiagonalisation, ONE hand-off per step
// Round 3's kernel (k_sytrd_dist, workgroup-wide: v_k out from the owner of column k, y back from everybody) spent 7 us per
// step at n = 900 on two hand-offs, five barriers and eleven polls in sequence.  Here every WAVE is autonomous and a step has
// one hand-off (4.9 ms instead of 6.4 at n = 900, 1.9 instead of 2.3 at n = 400; profiles/r4_dense_solver_timing.txt):
//   * columns are dealt cyclically to waves (column j -> wave j % (4 G)), whole columns in LDS, both triangles;
//   * with the y values of step k the owner of column k+1 also publishes that column as it stands (before update k).  Every
//     wave then forms w_k, the updated column a' = c - v w_{k+1} - w v_{k+1}, and from it v_{k+1}, tau_{k+1}, beta_{k+1}
//     REDUNDANTLY -- same instructions on the same words in every wave, so all copies agree bit for bit -- instead of waiting
//     for an owner to broadcast them;
//   * the rank-2 update of a wave's own columns with (v_k, w_k) and their products with v_{k+1} are one sweep, whose results
//     are the next step's published y.
// Vectors live in registers (lane l holds rows l, l + 64, ...), reductions are wave-wide shuffles, the only barrier of a step
// is the one behind the poll of the published words into LDS (double-buffered by step parity).  Same hand-off discipline as
// above (handoff.h): every word written once into a buffer pre-filled with an impossible pattern, bounded spins.
// pub: the records of k_sytrd_dist (v_k, tau_k, beta_k, y_k: what k_sytrd_back and the host read) followed by n - 1 records
// of n doubles, record k = column k (rows k ..) as published.
template <int RPL>
__device__ __forceinline__ double row_bcast(const double (&a)[RPL], int row) {
    double t = 0.0;
    const int ms = row >> 6;  // wave-uniform
#pragma unroll
    for (int m = 0; m < RPL; ++m)
        if (m == ms) t = a[m];
    return lane_bcast(t, row & 63);
}

template <int RPL, typename F, int... Is>
__device__ __forceinline__ void segments_impl(F& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int RPL, typename F>
__device__ __forceinline__ void segments(F& f) {
    segments_impl<RPL>(f, std::make_integer_sequence<int, RPL>{});
}

template <int RPL>
__global__ __launch_bounds__(kSyT) void k_sytrd_wave(int n, int G, int cw, int ldc, const double* __restrict__ A,
                                                    const double* __restrict__ diag_add, double* pub,
                                                    double* __restrict__ d_out, double* __restrict__ e_out, int* status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* cols = reinterpret_cast<double*>(smem_raw);       // [4 waves][cw][ldc], rows n .. ldc-1 zero
    double* land = cols + (size_t)4 * cw * ldc;                // [2 parities][y: ldc | c: ldc]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = 4 * G, gw = blockIdx.x * 4 + wave;
    const size_t S = 2 * (size_t)n + 2;
    double* cbase = pub + (size_t)(n - 2) * S;                 // column records, stride ldc
    double* mycols = cols + (size_t)wave * cw * ldc;
    const bool writer = gw == 0;

    // No per-lane predicate guards a memory operation in the step loop (every such `if` is an exec-mask branch, and ~360 of
    // them per step were most of a first version's 9 us): LDS rows are padded to 64 RPL and zero beyond n, rows that are
    // done with stay finite and meet a zero of v, w or v', and what must be masked is masked by a select on the VALUE.
    for (int i = tid; i < 4 * ldc; i += kSyT) land[i] = 0.0;
    for (int c = 0; c < cw; ++c) {  // lower triangle read, mirrored (SelfAdjointEigenSolver's convention, :207)
        const int j = gw + c * NW;
        for (int i = lane; i < ldc; i += 64) {
            double x = 0.0;
            if (j < n && i < n) {
                x = (i >= j) ? A[(size_t)j * n + i] : A[(size_t)i * n + j];
                if (i == j && diag_add != nullptr) x += diag_add[i];
            }
            mycols[(size_t)c * ldc + i] = x;
        }
    }
    // v_0, tau_0, beta_0 from column 0 (every wave reads it from the input)
    double v[RPL], w[RPL], vn[RPL];
    double tau, beta, dk;
    {
        double part = 0.0;
#pragma unroll
        for (int m = 0; m < RPL; ++m) {
            const int i = lane + 64 * m;
            double x = 0.0;
            if (i >= 1 && i < n) x = A[i];  // column 0, rows >= 1 (lower triangle)
            v[m] = x;
            part += i >= 2 ? x * x : 0.0;
        }
        part = wave_sum_dpp(part);
        const double alpha = row_bcast<RPL>(v, 1);
        dk = A[0] + (diag_add != nullptr ? diag_add[0] : 0.0);
        double scale = 0.0;
        tau = 0.0;
        beta = alpha;
        if (part != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + part), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
#pragma unroll
        for (int m = 0; m < RPL; ++m) {
            const int i = lane + 64 * m;
            v[m] = i == 1 ? 1.0 : (i >= 2 && i < n ? v[m] * scale : 0.0);
        }
    }
    __syncthreads();  // (land zeroed)
    // y_0 of the own columns (j >= 1), and column 1 as it stands
    for (int c = 0; c < cw; ++c) {
        const int j = gw + c * NW;
        if (j < 1 || j >= n) continue;  // wave-uniform
        const double* col = mycols + (size_t)c * ldc;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < RPL; ++m) acc += col[lane + 64 * m] * v[m];
        acc = wave_sum_dpp(acc);
        if (lane == 0) st_pub(pub + n + 2 + j, acc);
        if (j == 1) {
#pragma unroll
            for (int m = 0; m < RPL; ++m) st_pub(cbase + (size_t)1 * ldc + lane + 64 * m, col[lane + 64 * m]);
        }
    }
#ifdef NLE_SYTRD_PROBE
    unsigned long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, nspin = 0, t_a, t_b;
#endif
    // The step loop in RPL segments of (up to) 64 steps: in segment M0 the blocks of 64 rows below 64 M0 are done with, and
    // every row loop of the body starts at M0 AT COMPILE TIME -- half the instructions on average, and no branch (a run-time
    // skip of those blocks costs more than it saves, see above).  The row-(m < M0) entries of v, w, vn are never read.
    bool dead = false;  // a hand-off timed out (workgroup-uniform): the remaining segments do nothing
    auto segment = [&](auto M0c) __attribute__((always_inline)) {
        constexpr int M0 = decltype(M0c)::value;
        const int k_lo = max(0, 64 * M0 - 1), k_hi = dead ? -1 : min(n - 3, 64 * M0 + 62);
        for (int k = k_lo; k <= k_hi; ++k) {
#ifdef NLE_SYTRD_PROBE
            t_a = wall_clock64();
#endif
            // (1) the published y_k (j = k+1 ..) and column k+1 (rows k+1 ..) into LDS.  All of a thread's words are requested
            // together and re-requested until none is unset: a poll is a round trip to the coherent level of the memory system.
            double* ly = land + (size_t)(k & 1) * 2 * ldc;
            double* lc = ly + ldc;
            const double* yrec = pub + (size_t)k * S + n + 2;
            const double* crec = cbase + (size_t)(k + 1) * ldc;
            bool fail = false;
#ifdef NLE_SYTRD_PROBE
            unsigned spins_probe = 0;
#endif
            {
                // (straight-line rounds: a thread's words beyond n - 1 poll word n - 1 again instead of being guarded)
                constexpr int NP = (RPL + 3) / 4;
                u64 vy[NP], vc[NP];
                int idx[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    idx[q] = min(k + 1 + tid + kSyT * q, n - 1);
                    vy[q] = ld_pub(yrec + idx[q]);
                    vc[q] = ld_pub(crec + idx[q]);
                }
                // Up to 640 rows TWO sets of requests are in flight, half a round trip apart: a word that lands just after one
                // request passed is seen by the next ~0.3 us later instead of a whole round trip (~1 us) later (n = 200: 1.01 ->
                // 0.84 ms, n = 400: 2.07 -> 1.91).  Above, the doubled traffic of more pollers on more words eats the gain.
                constexpr bool DUAL = RPL <= 10;  // (measured again at n = 900 / 1152 with the probes: two sets there add 0.3 ms of poll time)
                unsigned spins = 0;
                u64 ny[NP], nc[NP];
                if constexpr (DUAL) {
                    __builtin_amdgcn_s_sleep(4);
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        ny[q] = ld_pub(yrec + idx[q]);
                        nc[q] = ld_pub(crec + idx[q]);
                    }
                }
                for (;;) {
                    bool missing = false;
#pragma unroll
                    for (int q = 0; q < NP; ++q) missing = missing || vy[q] == kUnset || vc[q] == kUnset;  // (waits for the older set only)
                    if (!missing) break;
                    if (++spins > kSpinLimit ||
                        ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        fail = true;
                        break;
                    }
                    if constexpr (!DUAL) __builtin_amdgcn_s_sleep(1);
                    // (DUAL: the younger set becomes the older one, merged into what is already known; a new set goes out)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const u64 oy = vy[q], oc = vc[q];
                        u64 fy, fc;
                        if constexpr (DUAL) {
                            fy = ny[q];
                            fc = nc[q];
                            ny[q] = ld_pub(yrec + idx[q]);
                            nc[q] = ld_pub(crec + idx[q]);
                        } else {
                            fy = ld_pub(yrec + idx[q]);
                            fc = ld_pub(crec + idx[q]);
                        }
                        vy[q] = oy == kUnset ? fy : oy;
                        vc[q] = oc == kUnset ? fc : oc;
                    }
                }
#ifdef NLE_SYTRD_PROBE
                spins_probe = spins;
#endif
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    ly[idx[q]] = __longlong_as_double((long long)vy[q]);
                    lc[idx[q]] = __longlong_as_double((long long)vc[q]);
                }
            }
#ifdef NLE_SYTRD_PROBE
            t_b = wall_clock64(); tp3 += t_b - t_a; t_a = t_b; nspin += spins_probe;
#endif
            if (__syncthreads_or(fail)) {
                if (tid == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                return;
            }
#ifdef NLE_SYTRD_PROBE
            t_b = wall_clock64(); tp0 += t_b - t_a; t_a = t_b;
#endif
            // (2) s = y . v, w = tau (y - (tau s / 2) v) on the active rows (i > k), zero elsewhere.  (Skipping the blocks of 64
            // rows that are done with costs more than it saves: 15 wave-uniform branches per loop, +1.3 us per step measured --
            // with one wave per SIMD nothing hides a branch's refetch.  The step loop is kept as straight as it can be.)
            double yl[RPL], cl[RPL];  // both LDS vectors requested up front
#pragma unroll
            for (int m = M0; m < RPL; ++m) {
                yl[m] = ly[lane + 64 * m];
                cl[m] = lc[lane + 64 * m];
            }
            const double yk1 = ly[k + 1];
            double pa = 0.0, pb = 0.0;  // two chains of dependent adds instead of one
#pragma unroll
            for (int m = M0; m < RPL; ++m) {
                const int i = lane + 64 * m;
                w[m] = (i > k && i < n) ? yl[m] : 0.0;
                if (m & 1) pb += w[m] * v[m];
                else pa += w[m] * v[m];
            }
            const double s = wave_sum_dpp(pa + pb);
            const double hs = 0.5 * tau * s;
#pragma unroll
            for (int m = M0; m < RPL; ++m) w[m] = tau * (w[m] - hs * v[m]);  // (both zero on the other rows)
            const double wk1 = tau * (yk1 - hs);  // w at row k+1, where v is 1
            // (3) column k+1 after update k, and from it v_{k+1}, tau_{k+1}, beta_{k+1}
            pa = 0.0;
            pb = 0.0;
#pragma unroll
            for (int m = M0; m < RPL; ++m) {
                const int i = lane + 64 * m;
                const double cv = cl[m] - __dadd_rn(__dmul_rn(v[m], wk1), w[m]);  // v_{k+1} of v_k is 1
                vn[m] = (i > k && i < n) ? cv : 0.0;
                const double sq = i > k + 2 ? vn[m] * vn[m] : 0.0;
                if (m & 1) pb += sq;
                else pa += sq;
            }
            const double part2 = wave_sum_dpp(pa + pb);
            const double dk1 = row_bcast<RPL>(vn, k + 1);
            const double alpha = row_bcast<RPL>(vn, k + 2);
            // (selects, not a branch: with part2 == 0 the quotients may be 0 / 0 and are discarded; reciprocals by v_rcp_f64 and
            // two Newton steps -- every wave computes the same bits, and H = I - tau v v^T only needs tau to an ulp or two)
            const bool nz = part2 != 0.0;
            const double bq = -copysign(sqrt(alpha * alpha + part2), alpha);
            double rb = __builtin_amdgcn_rcp(bq), rs = __builtin_amdgcn_rcp(alpha - bq);
            rb = fma(fma(-bq, rb, 1.0), rb, rb);
            rs = fma(fma(-(alpha - bq), rs, 1.0), rs, rs);
            rb = fma(fma(-bq, rb, 1.0), rb, rb);
            rs = fma(fma(-(alpha - bq), rs, 1.0), rs, rs);
            const double beta1 = nz ? bq : alpha;
            const double tau1 = nz ? (bq - alpha) * rb : 0.0;
            const double scale1 = nz ? rs : 0.0;
            const bool more = k + 3 < n;
            if (!more && writer && lane == 0) {  // the trailing 2 x 2 block: column n-2 after the last update is (d_{n-2}, e_{n-1})
                d_out[n - 2] = dk1;
                e_out[n - 1] = alpha;
            }
#pragma unroll
            for (int m = M0; m < RPL; ++m) {
                const int i = lane + 64 * m;
                vn[m] = i == k + 2 ? 1.0 : (i > k + 2 ? vn[m] * scale1 : 0.0);
            }
#ifdef NLE_SYTRD_PROBE
            t_b = wall_clock64(); tp1 += t_b - t_a; t_a = t_b;
#endif
            // (4) own columns j >= k+2: A -= v w^T + w v^T (products rounded separately: the two stored copies of an entry stay
            // bitwise equal), their products with v_{k+1} published as y_{k+1}; column k+2 published as it now stands
            double* yrec1 = pub + (size_t)(k + 1) * S + n + 2;
            double* crec1 = cbase + (size_t)(k + 2) * ldc;
            const int c0 = (k + 2 > gw) ? (k + 2 - gw + NW - 1) / NW : 0;
            for (int cc = c0; cc < cw; ++cc) {
                const int j = gw + cc * NW;
                if (j >= n) break;  // wave-uniform
                double* col = mycols + (size_t)cc * ldc;
                const double vj = row_bcast<RPL>(v, j), wj = row_bcast<RPL>(w, j);
                double x[RPL];  // all of the column's loads in flight together, then the arithmetic, then the stores
#pragma unroll
                for (int m = M0; m < RPL; ++m) x[m] = col[lane + 64 * m];
                double acc = 0.0;
#pragma unroll
                for (int m = M0; m < RPL; ++m) {
                    x[m] -= __dadd_rn(__dmul_rn(v[m], wj), __dmul_rn(w[m], vj));
                    acc += x[m] * vn[m];
                }
#pragma unroll
                for (int m = M0; m < RPL; ++m) col[lane + 64 * m] = x[m];
                if (more) {
                    if (j == k + 2) {
#pragma unroll
                        for (int m = M0; m < RPL; ++m) st_pub(crec1 + lane + 64 * m, x[m]);
                    }
                    acc = wave_sum_dpp(acc);
                    if (lane == 0) st_pub(yrec1 + j, acc);
                } else if (j == n - 1) {
                    const double dl = row_bcast<RPL>(x, n - 1);
                    if (lane == 0) d_out[n - 1] = dl;
                }
            }
#ifdef NLE_SYTRD_PROBE
            t_b = wall_clock64(); tp2 += t_b - t_a;
#endif
            if (writer) {  // the record the back-transformation and the host read (after this wave's part of the hand-off)
                double* rec = pub + (size_t)k * S;
#pragma unroll
                for (int m = M0; m < RPL; ++m) {
                    const int i = lane + 64 * m;
                    if (i > k && i < n) rec[i] = v[m];
                }
                if (lane == 0) {
                    rec[n] = tau;
                    rec[n + 1] = beta;
                    d_out[k] = dk;
                    e_out[k + 1] = beta;
                }
            }
#pragma unroll
            for (int m = M0; m < RPL; ++m) v[m] = vn[m];
            tau = tau1;
            beta = beta1;
            dk = dk1;
            }
    };
    segments<RPL>(segment);
    if (dead) return;
    if (writer && lane == 0) e_out[0] = 0.0;
#ifdef NLE_SYTRD_PROBE
    if (lane == 0 && (gw == 0 || gw == NW - 1)) printf("[sytrd probe] n=%d gw=%d: poll %.1f us (%llu extra rounds), barrier %.1f us, (2)(3) %.1f us, (4) %.1f us (wall clock, 100 MHz)\n", n, gw, tp3 * 0.01, nspin, tp0 * 0.01, tp1 * 0.01, tp2 * 0.01);
#endif
}

int sytrd_max_n() { return 1152; }

// Distribution of the one-hand-off form: four waves a workgroup, `cw` whole columns a wave.  Few workgroups mean few pollers
// of every published word, many mean a short sweep over the own columns on the critical path: measured (NLE_SYTRD_CW), four
// columns a wave up to n = 640 (n = 400: 25 workgroups), two above (n = 900: 113 workgroups, 5.9 against 6.6 ms with the
// upload; one column a wave is slower again).
static int sytrd_cols_per_wave(int n) {
    if (const char* e = std::getenv("NLE_SYTRD_CW")) return std::max(1, std::min(4, std::atoi(e)));
    return n > 640 ? 2 : 4;
}
// rows per lane the kernel is instantiated for, and with it the padded column length in LDS and in the published records
static int sytrd_rpl(int n) {
    const int r = (n + 63) / 64;
    for (int R : {5, 7, 10, 13, 15, 18})
        if (r <= R) return R;
    return 0;
}
int sytrd_groups(int n) {
    const int cw = sytrd_cols_per_wave(n);
    return (n + 4 * cw - 1) / (4 * cw);
}
size_t sytrd_pub_elems(int n) { return n > 2 ? (size_t)(n - 2) * (2 * (size_t)n + 2) + (size_t)n * 64 * sytrd_rpl(n) : 1; }

hipError_t sytrd_dist(hipStream_t s, int n, int G, const double* d_A, const double* d_diag_add, double* d_pub, double* d_d,
                      double* d_e, int* d_status) {
    if (n < 3 || n > sytrd_max_n()) return hipErrorInvalidValue;
    if (G <= 0) G = sytrd_groups(n);
    const int cw = (n + 4 * G - 1) / (4 * G);
    const int rpl = sytrd_rpl(n), ldc = 64 * rpl;
    size_t shm = (size_t)(4 * cw + 4) * ldc * 8;
    if (G > 240 || shm > 160 * 1024) return hipErrorInvalidValue;
    shm = std::max<size_t>(shm, 82 * 1024);  // more than half of a compute unit's LDS: one workgroup per compute unit
    hipError_t e = hipMemsetAsync(d_pub, 0xFF, sytrd_pub_elems(n) * sizeof(double), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_status, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
#define NLE_SYW(R)                                                                                                          \
    {                                                                                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_wave<R>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)shm);                                                                                  \
        if (e != hipSuccess) return e;                                                                                      \
        hipLaunchKernelGGL((k_sytrd_wave<R>), dim3(G), dim3(kSyT), shm, s, n, G, cw, ldc, d_A, d_diag_add, d_pub, d_d, d_e,  \
                           d_status);                                                                                       \
    }
    if (rpl == 5) NLE_SYW(5)
    else if (rpl == 7) NLE_SYW(7)
    else if (rpl == 10) NLE_SYW(10)
    else if (rpl == 13) NLE_SYW(13)
    else if (rpl == 15) NLE_SYW(15)
    else NLE_SYW(18)
#undef NLE_SYW
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ eigenvalues of T
// D[j], j = 0 .. n-1 DESCENDING.  Sturm count N(x) = number of eigenvalues < x = number of negative pivots of T - x I.
__global__ __launch_bounds__(256) void k_tridiag_bisect(int n, const double* __restrict__ d, const double* __restrict__ e,
                                                        double* __restrict__ D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* sd = reinterpret_cast<double*>(smem_raw);
    double* se2 = sd + n;
    double* sred = se2 + n;  // [4] lo, [4] hi, [4] max e^2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double glo = 1e300, ghi = -1e300, emax = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double di = d[i], el = i > 0 ? e[i] : 0.0, er = i + 1 < n ? e[i + 1] : 0.0;
        sd[i] = di;
        se2[i] = el * el;
        const double rad = fabs(el) + fabs(er);
        glo = fmin(glo, di - rad);
        ghi = fmax(ghi, di + rad);
        emax = fmax(emax, el * el);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        glo = fmin(glo, __shfl_xor(glo, off));
        ghi = fmax(ghi, __shfl_xor(ghi, off));
        emax = fmax(emax, __shfl_xor(emax, off));
    }
    if (lane == 0) {
        sred[wave] = glo;
        sred[4 + wave] = ghi;
        sred[8 + wave] = emax;
    }
    __syncthreads();
    glo = fmin(fmin(sred[0], sred[1]), fmin(sred[2], sred[3]));
    ghi = fmax(fmax(sred[4], sred[5]), fmax(sred[6], sred[7]));
    emax = fmax(fmax(sred[8], sred[9]), fmax(sred[10], sred[11]));
    const double tnorm = fmax(fabs(glo), fabs(ghi));
    const double ulp = 2.220446049250313e-16;
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax);
    const double atol = ulp * tnorm;
    glo -= 2.0 * ulp * tnorm * n + 2.0 * pivmin;
    ghi += 2.0 * ulp * tnorm * n + 2.0 * pivmin;
    const int j = blockIdx.x * 4 + wave;
    if (j >= n) return;
    const int a = n - 1 - j;  // ascending index: N(lo) <= a < N(hi)
    double lo = glo, hi = ghi;
    for (int round = 0; round < 24; ++round) {
        const double width = hi - lo;
        if (width <= fmax(atol, 2.0 * ulp * fmax(fabs(lo), fabs(hi)))) break;
        const double x = lo + width * ((double)(lane + 1) * (1.0 / 65.0));
        double q = sd[0] - x;
        if (fabs(q) < pivmin) q = -pivmin;
        int cnt = q < 0.0 ? 1 : 0;
        for (int i = 1; i < n; ++i) {
            // e^2 / q with v_rcp_f64 + two Newton steps (1 ulp, 6 instructions against the 13 of the IEEE sequence: this
            // recurrence is one dependent chain of n steps per round)
            double rq = __builtin_amdgcn_rcp(q);
            rq = fma(fma(-q, rq, 1.0), rq, rq);
            rq = fma(fma(-q, rq, 1.0), rq, rq);
            q = fma(-se2[i], rq, sd[i] - x);
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0 ? 1 : 0;
        }
        const u64 mask = __ballot(cnt >= a + 1);
        if (mask == 0) {
            lo = __shfl(x, 63);
        } else {
            const int f = __ffsll((long long)mask) - 1;
            const double xf = __shfl(x, f), xp = __shfl(x, f > 0 ? f - 1 : 0);
            hi = xf;
            if (f > 0) lo = xp;
        }
    }
    if (lane == 0) D[j] = 0.5 * (lo + hi);
}

hipError_t tridiag_bisect(hipStream_t s, int n, const double* d_d, const double* d_e, double* d_D) {
    if (n < 1) return hipErrorInvalidValue;
    const size_t shm = ((size_t)2 * n + 12) * sizeof(double);
    hipLaunchKernelGGL(k_tridiag_bisect, dim3((n + 3) / 4), dim3(256), shm, s, n, d_d, d_e, d_D);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ back-transformation
// Z (n x K, column stride ldz) <- Q Z,  Q = H_0 .. H_{n-3} from the reduction's records
template <int NM>
__global__ __launch_bounds__(256) void k_sytrd_back(int n, const double* __restrict__ pub, int K, double* __restrict__ Z,
                                                    int ldz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= K) return;
    const size_t S = 2 * (size_t)n + 2;
    double z[NM], v[NM], vn[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int i = lane + 64 * m;
        z[m] = i < n ? Z[(size_t)c * ldz + i] : 0.0;
    }
    // The reflectors are applied in NM segments of (up to) 64 steps, last rows first: in segment M0 (steps k with
    // (k + 1) >> 6 == M0) the reflector is zero above row 64 M0, so the row loops start at M0 at compile time (the prefetch of
    // the next record at M0 - 1: it may belong to the next segment).  No run-time skip, no branch per block (see k_sytrd_wave).
    double tau = 0.0, taun = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) v[m] = vn[m] = 0.0;
    {
        const double* rec = pub + (size_t)(n - 3) * S;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int i = lane + 64 * m;
            v[m] = (i > n - 3 && i < n) ? rec[i] : 0.0;
        }
        tau = rec[n];
    }
    auto segment = [&](auto M0c) __attribute__((always_inline)) {
        constexpr int M0 = NM - 1 - decltype(M0c)::value;  // descending
        constexpr int ML = M0 > 0 ? M0 - 1 : 0;
        const int k_hi = min(n - 3, 64 * M0 + 62), k_lo = max(0, 64 * M0 - 1);
        for (int k = k_hi; k >= k_lo; --k) {
            if (k > 0) {
                const double* rec = pub + (size_t)(k - 1) * S;
#pragma unroll
                for (int m = ML; m < NM; ++m) {
                    const int i = lane + 64 * m;
                    vn[m] = (i > k - 1 && i < n) ? rec[i] : 0.0;
                }
                taun = rec[n];
            }
            double dot = 0.0;
#pragma unroll
            for (int m = M0; m < NM; ++m) dot += v[m] * z[m];
            dot = wave_sum(dot) * tau;
#pragma unroll
            for (int m = M0; m < NM; ++m) z[m] -= dot * v[m];
#pragma unroll
            for (int m = ML; m < NM; ++m) v[m] = vn[m];
            tau = taun;
        }
    };
    segments<NM>(segment);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int i = lane + 64 * m;
        if (i < n) Z[(size_t)c * ldz + i] = z[m];
    }
}

hipError_t sytrd_back(hipStream_t s, int n, const double* d_pub, int K, double* d_Z, int ldz) {
    if (n < 3 || n > sytrd_max_n() || K < 1) return hipErrorInvalidValue;
    const dim3 grid((K + 3) / 4), block(256);
    if (n <= 256)
        hipLaunchKernelGGL((k_sytrd_back<4>), grid, block, 0, s, n, d_pub, K, d_Z, ldz);
    else if (n <= 512)
        hipLaunchKernelGGL((k_sytrd_back<8>), grid, block, 0, s, n, d_pub, K, d_Z, ldz);
    else if (n <= 832)
        hipLaunchKernelGGL((k_sytrd_back<13>), grid, block, 0, s, n, d_pub, K, d_Z, ldz);
    else
        hipLaunchKernelGGL((k_sytrd_back<18>), grid, block, 0, s, n, d_pub, K, d_Z, ldz);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ Cholesky + inverse
constexpr int kChB = 32;  // panel width

// Panel [j0, j0 + nb) of the factorisation (A: n x n column-major, lower triangle; columns < j0 already hold L and the
// trailing block its Schur complement).  Every workgroup factors the diagonal block in LDS (identical arithmetic) --
// which is why NOBODY may write that block back in place: a workgroup that starts late would read a factored block and
// factor it again (seen under load: the launch shares the chip with the Gram kernels).  The factored block goes to
// `diag` (32 x 32 column-major per panel) and is put in place by k_potrf_finish; workgroup b solves rows j0 + nb + 256 b + tid
// of the panel (one row per thread); workgroup 0 also writes the block's inverse into Linv.  status |= 2 on a
// non-positive pivot (the factor is then meaningless; the pivot is replaced by 1 so that nothing overflows).
__device__ __forceinline__ double bcast_lane(double v, int src) {  // v of lane `src` (a compile-time constant) to every lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void k_potrf_panel(int n, int j0, double* __restrict__ A, double* __restrict__ diag,
                                                     double* __restrict__ Linv, int* status) {
    __shared__ double sL[kChB][kChB + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nb = min(kChB, n - j0);
    if (tid < 64) {
        // wave 0: lane r holds row r of the diagonal block in registers (rows >= nb: identity); right-looking
        // factorisation with every exchange a lane broadcast -- no LDS round trips, no barriers
        const int r = lane & 31;
        double a[kChB];
#pragma unroll
        for (int c = 0; c < kChB; ++c) {
            double v = (r == c) ? 1.0 : 0.0;
            if (r < nb && c < nb && r >= c) v = A[(size_t)(j0 + c) * n + j0 + r];
            a[c] = v;
        }
        bool bad = false;
#pragma unroll
        for (int c = 0; c < kChB; ++c) {
            double d = bcast_lane(a[c], c);
            if (!(d > 0.0)) {
                bad = true;
                d = 1.0;
            }
            const double piv = sqrt(d), rp = 1.0 / piv;
            a[c] = (r == c) ? piv : a[c] * rp;
#pragma unroll
            for (int j = c + 1; j < kChB; ++j) {
                const double ljc = bcast_lane(a[c], j);
                a[j] -= a[c] * ljc;  // meaningful for r >= j
            }
        }
        if (lane < 32) {
#pragma unroll
            for (int c = 0; c < kChB; ++c) sL[r][c] = (c <= r) ? a[c] : 0.0;
        }
        if (blockIdx.x == 0) {
            if (bad && lane == 0) atomicOr(status, 2);
            double* dg = diag + (size_t)(j0 / kChB) * kChB * kChB;
            if (lane < 32) {
#pragma unroll
                for (int c = 0; c < kChB; ++c) dg[(size_t)c * kChB + r] = (c <= r && r < nb) ? a[c] : 0.0;
            }
            // column `r` of the block's inverse: x_i = (delta_ir - sum_{k < i} L(i,k) x_k) / L(i,i); x_k = 0 for k < r
            double x[kChB];
#pragma unroll
            for (int i = 0; i < kChB; ++i) {
                double acc = (i == r) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) acc -= bcast_lane(a[k], i) * x[k];
                x[i] = acc / bcast_lane(a[i], i);
            }
            if (lane < nb) {
#pragma unroll
                for (int i = 0; i < kChB; ++i)
                    if (i < nb) Linv[(size_t)(j0 + r) * n + j0 + i] = x[i];
            }
        }
    }
    __syncthreads();
    const int row = j0 + nb + blockIdx.x * 256 + tid;
    if (row < n) {  // x L11^T = a
        double x[kChB];
#pragma unroll
        for (int c = 0; c < kChB; ++c) x[c] = c < nb ? A[(size_t)(j0 + c) * n + row] : 0.0;
#pragma unroll
        for (int c = 0; c < kChB; ++c) {
            if (c < nb) {
                double acc = x[c];
#pragma unroll
                for (int k = 0; k < kChB; ++k)
                    if (k < c) acc -= x[k] * sL[c][k];
                x[c] = acc / sL[c][c];
            }
        }
#pragma unroll
        for (int c = 0; c < kChB; ++c)
            if (c < nb) A[(size_t)(j0 + c) * n + row] = x[c];
    }
}

// the factored diagonal blocks into place, zeros above the diagonal
__global__ void k_potrf_finish(double* __restrict__ A, int n, const double* __restrict__ diag) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < (size_t)n * n; t += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(t % n), c = (int)(t / n);
        if (r < c) {
            A[t] = 0.0;
        } else if (r / kChB == c / kChB) {
            A[t] = diag[(size_t)(c / kChB) * kChB * kChB + (size_t)(c % kChB) * kChB + (r % kChB)];
        }
    }
}

__global__ void k_fill64(double* p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// sum of squares of n doubles: 128 workgroup partials, then one wave adds them in a fixed order
__global__ __launch_bounds__(256) void k_sumsq_part(const double* __restrict__ x, size_t n, double* __restrict__ part) {
    __shared__ double sred[4];
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}
__global__ void k_sum128(const double* __restrict__ part, double* out) {
    double v = part[threadIdx.x] + part[threadIdx.x + 64];
    v = wave_sum(v);
    if (threadIdx.x == 0) out[0] = v;
}

// One level of the inverse of a lower-triangular matrix by doubling: with the diagonal blocks of size s already inverted
// (X11, X22 in place in X), pair b = blockIdx.y gets X21 = -X22 L21 X11 in two phases
//   phase 0: T_b (m x s) = L21 X11        phase 1: X21 = -X22 T_b        (rows row0 = 2 b s + s .. , m = min(s, n - row0))
// one wave per 16 x 16 output tile on v_mfma_f64_16x16x4_f64, operands straight from L2 (the matrices are a few MB).
__global__ __launch_bounds__(256) void k_trtri_level(int phase, int n, int s, const double* __restrict__ L,
                                                     double* __restrict__ X, double* __restrict__ T) {
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y, col0 = 2 * b * s, row0 = col0 + s;
    const int m = min(s, n - row0);
    if (m <= 0) return;
    const int rt = (m + 15) / 16, ct = s / 16;
    const int t = blockIdx.x * 4 + wave;
    if (t >= rt * ct) return;  // wave-uniform
    const int ti = t / ct, tj = t - ti * ct;
    const int ai = ti * 16 + l15, bj = tj * 16 + l15;
    const bool aok = ai < m;
    double* Tb = T + (size_t)b * s * s;  // m x s, column-major, leading dimension s
    const int kk = phase == 0 ? s : m;
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < kk; k0 += 4) {
        const int k = k0 + kq;
        double av = 0.0, bv = 0.0;
        if (k < kk) {
            if (phase == 0) {
                if (aok) av = L[(size_t)(col0 + k) * n + row0 + ai];   // L21(ai, k)
                bv = X[(size_t)(col0 + bj) * n + col0 + k];            // X11(k, bj)
            } else {
                if (aok) av = X[(size_t)(row0 + k) * n + row0 + ai];   // X22(ai, k)
                bv = Tb[(size_t)bj * s + k];                           // T(k, bj)
            }
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ro = ti * 16 + kq + 4 * e;
        if (ro < m) {
            if (phase == 0) Tb[(size_t)bj * s + ro] = acc[e];
            else X[(size_t)(col0 + bj) * n + row0 + ro] = -acc[e];
        }
    }
}

// dst (n x n, column stride ldd) = transpose of src (n x n, column stride lds)
__global__ void k_transpose64(int n, const double* __restrict__ src, int lds, double* __restrict__ dst, int ldd) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
    for (int r = ty; r < 32; r += 8)
        if (bx + tx < n && by + r < n) tile[r][tx] = src[(size_t)(by + r) * lds + bx + tx];  // element (row bx+tx, col by+r)
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (by + tx < n && bx + r < n) dst[(size_t)(bx + r) * ldd + by + tx] = tile[tx][r];  // dst(row by+tx, col bx+r)
}
// dst = the symmetric matrix whose lower triangle is src's (what SelfAdjointEigenSolver sees, src/filter.cpp:207)
__global__ void k_symm_lower64(int n, const double* __restrict__ src, double* __restrict__ dst) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < (size_t)n * n; t += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(t % n), c = (int)(t / n);
        dst[t] = r >= c ? src[t] : src[(size_t)r * n + c];
    }
}
hipError_t transpose64(hipStream_t s, int n, const double* d_src, double* d_dst, int lds, int ldd) {
    hipLaunchKernelGGL(k_transpose64, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, s, n, d_src, lds > 0 ? lds : n, d_dst,
                       ldd > 0 ? ldd : n);
    return hipGetLastError();
}
hipError_t symm_lower64(hipStream_t s, int n, const double* d_src, double* d_dst) {
    hipLaunchKernelGGL(k_symm_lower64, dim3(256), dim3(256), 0, s, n, d_src, d_dst);
    return hipGetLastError();
}
// dst (n x n, column-major) = diag(dl) src[:n, :n] diag(dr), src with column stride lds
__global__ void k_scale_rc64(int n, const double* __restrict__ src, int lds, const double* __restrict__ dl,
                             const double* __restrict__ dr, double* __restrict__ dst) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < (size_t)n * n; t += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(t % n), c = (int)(t / n);
        dst[t] = dl[r] * src[(size_t)c * lds + r] * dr[c];
    }
}
hipError_t scale_rc64(hipStream_t s, int n, const double* d_src, int lds, const double* d_dl, const double* d_dr, double* d_dst) {
    hipLaunchKernelGGL(k_scale_rc64, dim3(256), dim3(256), 0, s, n, d_src, lds, d_dl, d_dr, d_dst);
    return hipGetLastError();
}
hipError_t fill64(hipStream_t s, double* d_p, size_t n, double v) {
    hipLaunchKernelGGL(k_fill64, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, d_p, n, v);
    return hipGetLastError();
}

// d_A (n x n column-major, lower triangle read) -> d_L = its Cholesky factor (lower, zeros above the diagonal), d_Linv =
// L^-1 (likewise), d_scal[0] = trace(A^-1) = ||L^-1||_F^2; *d_status |= 2 if A is not positive definite.  d_A is left
// untouched; d_tmp: n x 32 doubles of scratch; d_minus: 32 doubles (filled here with -1).
// scratch of potrf_inverse: the doubling levels' T blocks (pairs x s x s at level s), the factored diagonal blocks,
// 32 x (-1), 128 partial sums
static size_t trtri_T_elems(int n) {
    size_t mx = 1;
    for (long long sz = kChB; sz < n; sz *= 2) {
        const size_t pairs = (size_t)((n + 2 * sz - 1) / (2 * sz));
        mx = std::max(mx, pairs * (size_t)sz * (size_t)sz);
    }
    return mx;
}
size_t potrf_tmp_elems(int n) { return trtri_T_elems(n) + (size_t)kChB * ((size_t)n + kChB) + kChB + 128; }

hipError_t potrf_inverse(hipStream_t s, int n, const double* d_A, double* d_L, double* d_Linv, double* d_tmp, double* d_scal,
                         int* d_status) {
    if (n < 1) return hipErrorInvalidValue;
    double* d_T = d_tmp;                                        // pairs x s x s at level s
    double* d_diag = d_tmp + trtri_T_elems(n);                  // one 32 x 32 block per panel
    double* d_minus = d_diag + (size_t)kChB * (n + kChB);       // 32 x (-1)
    double* d_part = d_minus + kChB;                            // 128 partial sums
    hipError_t e = hipMemcpyAsync(d_L, d_A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_Linv, 0, (size_t)n * n * sizeof(double), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fill64, dim3(1), dim3(64), 0, s, d_minus, (size_t)kChB, -1.0);
    for (int j0 = 0; j0 < n; j0 += kChB) {
        const int nb = std::min(kChB, n - j0), below = n - j0 - nb;
        hipLaunchKernelGGL(k_potrf_panel, dim3(std::max(1, (below + 255) / 256)), dim3(256), 0, s, n, j0, d_L, d_diag, d_Linv,
                           d_status);
        if (below > 0) {  // A22 -= L21 L21^T (both triangles; only the lower one is read later)
            double* a22 = d_L + (size_t)(j0 + nb) * n + j0 + nb;
            const double* l21 = d_L + (size_t)j0 * n + j0 + nb;
            // (the update is symmetric: written through the transposed strides, so that the 16 lanes of a tile column store
            // to consecutive addresses instead of 16 different columns)
            e = gemm64s(s, below, below, nb, l21, 1, n, l21, n, 1, a22, n, 1, nullptr, d_minus, nullptr, a22, n, 1);
            if (e != hipSuccess) return e;
        }
    }
    hipLaunchKernelGGL(k_potrf_finish, dim3(256), dim3(256), 0, s, d_L, n, d_diag);
    // L^-1 by doubling: the 32 x 32 diagonal blocks are inverted (k_potrf_panel); level s joins pairs of s x s blocks
    for (int sz = kChB; sz < n; sz *= 2) {
        const int pairs = (n + 2 * sz - 1) / (2 * sz);
        const dim3 grid((unsigned)(((sz / 16) * (sz / 16) + 3) / 4), (unsigned)pairs);
        hipLaunchKernelGGL(k_trtri_level, grid, dim3(256), 0, s, 0, n, sz, d_L, d_Linv, d_T);
        hipLaunchKernelGGL(k_trtri_level, grid, dim3(256), 0, s, 1, n, sz, d_L, d_Linv, d_T);
    }
    hipLaunchKernelGGL(k_sumsq_part, dim3(128), dim3(256), 0, s, d_Linv, (size_t)n * n, d_part);
    hipLaunchKernelGGL(k_sum128, dim3(1), dim3(64), 0, s, d_part, d_scal);
    return hipGetLastError();
}

}  // namespace nlek
