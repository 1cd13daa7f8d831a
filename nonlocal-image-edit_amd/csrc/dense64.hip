// Dense fp64 algebra of order n <= 1152 on the device: what is left of `eigenDecomposition` (reference src/filter.cpp:204-228,
// called at :262, :287, :313) and of the Cholesky shortcuts once everything N-sized is fast -- at p = 400..900 samples the two
// symmetric eigen-computations and the factorisation of K_A were half of a train step on one host core.
//
//   sytrd_dist        Householder tridiagonalisation, ONE persistent launch of G workgroups.  The matrix lives in LDS, whole
//                     columns dealt cyclically (column j -> workgroup j % G, both triangles stored, so y = A v needs no
//                     cross-workgroup sum: a workgroup forms the entries of y that belong to its own columns).  Per step k two
//                     hand-offs through global memory, both "the data is the flag" (MI355X_MICROARCH.md, visibility: 8-byte
//                     agent-scope atomic stores and loads on both sides, every word written once per launch into a buffer
//                     pre-filled with an impossible bit pattern, consumers re-read until no word is unset):
//                       v_k, tau_k  from the owner of column k to everybody      (one -> all)
//                       y_j         from the owner of column j to everybody      (all -> all)
//                     The owner of column k+1 updates that column first, forms v_{k+1} and publishes it before it touches its
//                     other columns, so the next step's vector is in flight under the rank-2 update.
//   tridiag_bisect    all eigenvalues of the tridiagonal matrix by Sturm counts, one wave per eigenvalue, 64 section points
//                     per round (the interval shrinks 65x per pass over d, e): ~10 rounds.
//   sytrd_back        back-transformation of K eigenvectors of T (one wave per vector, the vector in registers).
//   potrf / trtri     blocked Cholesky factor and its inverse: 32-column panels (one kernel: every workgroup re-factors the
//                     32 x 32 diagonal block in LDS and solves its own rows of the panel), trailing update and the inverse's
//                     block rows on k_gemm64s (fp64 MFMA).
// The eigenvectors of T for the wanted eigenvalues come from the host's inverse iteration (eigen_sym.cpp, O(n k)).
#include "kernels.h"

#include <algorithm>

namespace nlek {

namespace {

typedef unsigned long long u64;
constexpr u64 kUnset = ~0ull;  // a NaN no arithmetic produces (hardware NaNs are 0x7FF8.. / 0xFFF8..)
constexpr int kSyT = 256;      // threads per workgroup of the reduction
constexpr int kSyRows = 5;     // rows per thread: n <= 1152 < 5 * 256
constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ u64 ld_pub(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_pub(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// re-read one published word until it is set; gives up (fail = true) after kSpinLimit polls or when another workgroup
// has flagged the launch as failed
__device__ __forceinline__ double wait_pub(const double* p, const int* status, bool& fail) {
    u64 b;
    unsigned spins = 0;
    while ((b = ld_pub(p)) == kUnset) {
        if (++spins > kSpinLimit) {
            fail = true;
            break;
        }
        if ((spins & 1023u) == 0 &&
            __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            fail = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ tridiagonalisation
// pub: (n - 2) records of 2 n + 2 doubles, one per step k: v_k (entries k+1 .. n-1, v_k[k+1] = 1), tau_k at [n], beta_k at
// [n + 1], then y (entries k+1 .. n-1) from [n + 2].  H_k = I - tau_k v_k v_k^T; T = Q^T A Q, Q = H_0 H_1 .. H_{n-3};
// d_out[i] = T(i, i), e_out[i] = T(i, i-1) (e_out[0] = 0): the layout eigen_sym.cpp's tridiagonal routines take.
size_t sytrd_pub_elems(int n) { return n > 2 ? (size_t)(n - 2) * (2 * (size_t)n + 2) : 1; }

__global__ __launch_bounds__(kSyT) void k_sytrd_dist(int n, int G, int ldp, const double* __restrict__ A,
                                                    const double* __restrict__ diag_add, double* pub,
                                                    double* __restrict__ d_out, double* __restrict__ e_out, int* status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int cloc = (n + G - 1) / G;
    double* panel = reinterpret_cast<double*>(smem_raw);  // [cloc][ldp]: column g + l G, all rows
    double* sv0 = panel + (size_t)cloc * ldp;             // v_k, double-buffered by step parity
    double* sv1 = sv0 + ldp;
    double* sw = sv1 + ldp;                               // w_k
    double* sred = sw + ldp;                              // [0..3] norm partials, [4..7] y.v partials
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t S = 2 * (size_t)n + 2;

    for (int l = 0; l < cloc; ++l) {  // lower triangle read, mirrored (SelfAdjointEigenSolver's convention, :207)
        const int j = g + l * G;
        if (j >= n) break;
        for (int i = tid; i < n; i += kSyT) {
            double v = (i >= j) ? A[(size_t)j * n + i] : A[(size_t)i * n + j];
            if (i == j && diag_add != nullptr) v += diag_add[i];
            panel[(size_t)l * ldp + i] = v;
        }
    }
    __syncthreads();

    // v_k, tau_k, beta_k from column k (local column l) as it stands; published, and d_k, e_{k+1} written
    auto householder = [&](int k, int l) {
        const double* col = panel + (size_t)l * ldp;
        double part = 0.0;
        for (int i = k + 2 + tid; i < n; i += kSyT) part += col[i] * col[i];
        part = wave_sum(part);
        if (lane == 0) sred[wave] = part;
        __syncthreads();
        const double xn2 = (sred[0] + sred[1]) + (sred[2] + sred[3]);
        const double alpha = col[k + 1];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xn2 != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        double* rec = pub + (size_t)k * S;
        for (int i = k + 2 + tid; i < n; i += kSyT) st_pub(rec + i, col[i] * scale);
        if (tid == 0) {
            st_pub(rec + k + 1, 1.0);
            st_pub(rec + n, tau);
            st_pub(rec + n + 1, beta);
            d_out[k] = col[k];
            e_out[k + 1] = beta;
        }
    };

    if (g == 0) householder(0, 0);
    for (int k = 0; k + 2 < n; ++k) {
        const double* rec = pub + (size_t)k * S;
        double* sv = (k & 1) ? sv1 : sv0;
        bool fail = false;
        double vr[kSyRows], wr[kSyRows];
        // (1) v_k and tau_k
#pragma unroll
        for (int m = 0; m < kSyRows; ++m) {
            const int i = k + 1 + tid + kSyT * m;
            vr[m] = 0.0;
            if (i < n) {
                vr[m] = wait_pub(rec + i, status, fail);
                sv[i] = vr[m];
            }
        }
        const double tau = wait_pub(rec + n, status, fail);
        if (__syncthreads_or(fail)) {  // (also: every thread is done with the previous step's update of the panel)
            if (tid == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // (2) y_j = A(:, j) . v for this workgroup's columns j > k, published
        double* yrec = pub + (size_t)k * S + n + 2;
        for (int l = wave; l < cloc; l += 4) {
            const int j = g + l * G;
            if (j <= k || j >= n) continue;
            const double* col = panel + (size_t)l * ldp;
            double acc = 0.0;
            for (int i = k + 1 + lane; i < n; i += 64) acc += col[i] * sv[i];
            acc = wave_sum(acc);
            if (lane == 0) st_pub(yrec + j, acc);
        }
        // (3) all of y; s = y . v; w = tau (y - (tau s / 2) v)
        double part = 0.0;
#pragma unroll
        for (int m = 0; m < kSyRows; ++m) {
            const int i = k + 1 + tid + kSyT * m;
            wr[m] = 0.0;
            if (i < n) {
                wr[m] = wait_pub(yrec + i, status, fail);
                part += wr[m] * vr[m];
            }
        }
        part = wave_sum(part);
        if (lane == 0) sred[4 + wave] = part;
        if (__syncthreads_or(fail)) {
            if (tid == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const double s = (sred[4] + sred[5]) + (sred[6] + sred[7]);
        const double hs = 0.5 * tau * s;
#pragma unroll
        for (int m = 0; m < kSyRows; ++m) {
            const int i = k + 1 + tid + kSyT * m;
            if (i < n) {
                wr[m] = tau * (wr[m] - hs * vr[m]);
                sw[i] = wr[m];
            }
        }
        __syncthreads();
        // (4) A -= v w^T + w v^T on rows and columns > k.  Products rounded separately (no fused multiply-add), so that the
        // two stored copies of an entry, A(i, j) here and A(j, i) in column i's workgroup, stay bitwise equal.
        auto update_col = [&](int l) {
            const int j = g + l * G;
            double* col = panel + (size_t)l * ldp;
            const double vj = sv[j], wj = sw[j];
#pragma unroll
            for (int m = 0; m < kSyRows; ++m) {
                const int i = k + 1 + tid + kSyT * m;
                if (i < n) col[i] -= __dadd_rn(__dmul_rn(vr[m], wj), __dmul_rn(wr[m], vj));
            }
        };
        const bool next_owner = (k + 3 < n) && ((k + 1) % G == g);  // column k+1 is reduced in step k+1 <= n-3
        const int l1 = (k + 1) / G;
        if (next_owner) {
            if (tau != 0.0) update_col(l1);
            __syncthreads();
            householder(k + 1, l1);
        }
        if (tau != 0.0)
            for (int l = 0; l < cloc; ++l) {
                const int j = g + l * G;
                if (j <= k || j >= n || (next_owner && l == l1)) continue;
                update_col(l);
            }
    }
    __syncthreads();
    // the trailing 2 x 2 block
    if (tid == 0) {
        if ((n - 2) % G == g) {
            const double* col = panel + (size_t)((n - 2) / G) * ldp;
            d_out[n - 2] = col[n - 2];
            e_out[n - 1] = col[n - 1];
        }
        if ((n - 1) % G == g) d_out[n - 1] = panel[(size_t)((n - 1) / G) * ldp + n - 1];
        if (g == 0) e_out[0] = 0.0;
    }
}

int sytrd_max_n() { return 1152; }

// number of workgroups.  Measured (profiles/r3_dense_solver_timing.txt): the rank-2 update of a workgroup's own columns is a
// third of a step, so more workgroups win until the all-to-all hand-off of y grows: 32 below n = 256, 64 up to 832, 128
// above (also what the LDS then allows).
int sytrd_groups(int n) {
    const int want = n < 256 ? 32 : (n <= 832 ? 64 : 128);
    for (int G : {16, 32, 64, 128}) {
        if (G < want) continue;
        const size_t cloc = (size_t)(n + G - 1) / G;
        const size_t bytes = (cloc + 3) * (size_t)((n + 1) & ~1) * 8 + 64;
        if (bytes <= 150 * 1024) return G;
    }
    return 0;
}

hipError_t sytrd_dist(hipStream_t s, int n, int G, const double* d_A, const double* d_diag_add, double* d_pub, double* d_d,
                      double* d_e, int* d_status) {
    if (n < 3 || n > sytrd_max_n()) return hipErrorInvalidValue;
    if (G <= 0) G = sytrd_groups(n);
    const int ldp = (n + 1) & ~1;
    const size_t cloc = (size_t)(n + G - 1) / G;
    size_t shm = (cloc + 3) * (size_t)ldp * 8 + 64;
    if (G <= 0 || G > 128 || shm > 160 * 1024) return hipErrorInvalidValue;
    shm = std::max<size_t>(shm, 82 * 1024);  // more than half of a compute unit's LDS: one workgroup per compute unit
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sytrd_dist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shm);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_pub, 0xFF, sytrd_pub_elems(n) * sizeof(double), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_status, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sytrd_dist, dim3(G), dim3(kSyT), shm, s, n, G, ldp, d_A, d_diag_add, d_pub, d_d, d_e, d_status);
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
            q = (sd[i] - x) - se2[i] / q;
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
    auto load = [&](int k, double (&dst)[NM], double& tau) {
        const double* rec = pub + (size_t)k * S;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int i = lane + 64 * m;
            dst[m] = (i > k && i < n) ? rec[i] : 0.0;
        }
        tau = rec[n];
    };
    double tau, taun = 0.0;
    load(n - 3, v, tau);
    for (int k = n - 3; k >= 0; --k) {
        if (k > 0) load(k - 1, vn, taun);
        double dot = 0.0;
#pragma unroll
        for (int m = 0; m < NM; ++m) dot += v[m] * z[m];
        dot = wave_sum(dot) * tau;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            z[m] -= dot * v[m];
            v[m] = vn[m];
        }
        tau = taun;
    }
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
__global__ __launch_bounds__(256) void k_potrf_panel(int n, int j0, double* __restrict__ A, double* __restrict__ diag,
                                                     double* __restrict__ Linv, int* status) {
    __shared__ double sL[kChB][kChB + 1];
    __shared__ double sX[kChB][kChB + 1];
    const int tid = threadIdx.x;
    const int nb = min(kChB, n - j0);
    for (int t = tid; t < kChB * kChB; t += 256) {
        const int r = t % kChB, c = t / kChB;
        sL[r][c] = (r < nb && c < nb && r >= c) ? A[(size_t)(j0 + c) * n + j0 + r] : 0.0;
    }
    __syncthreads();
    for (int c = 0; c < nb; ++c) {
        double dcc = sL[c][c];
        __syncthreads();
        if (!(dcc > 0.0)) {
            if (tid == 0 && blockIdx.x == 0) atomicOr(status, 2);
            dcc = 1.0;
        }
        const double piv = sqrt(dcc), rp = 1.0 / piv;
        if (tid < nb && tid >= c) sL[tid][c] = (tid == c) ? piv : sL[tid][c] * rp;
        __syncthreads();
        // trailing block of the diagonal block: (r, cc) with cc > c, r >= cc
        for (int t = tid; t < kChB * kChB; t += 256) {
            const int r = t % kChB, cc = t / kChB;
            if (cc > c && r >= cc && r < nb) sL[r][cc] -= sL[r][c] * sL[cc][c];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        double* dg = diag + (size_t)(j0 / kChB) * kChB * kChB;
        for (int t = tid; t < kChB * kChB; t += 256) {
            const int r = t % kChB, c = t / kChB;
            dg[t] = (r < nb && c < nb && r >= c) ? sL[r][c] : 0.0;
        }
        if (tid < nb) {  // column tid of the block's inverse by forward substitution
            const int c = tid;
            for (int i = 0; i < nb; ++i) {
                double acc = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) acc -= sL[i][k] * sX[k][c];
                sX[i][c] = i >= c ? acc / sL[i][i] : 0.0;
            }
            for (int i = 0; i < nb; ++i) Linv[(size_t)(j0 + c) * n + j0 + i] = sX[i][c];
        }
    }
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
// out[0] = sum of squares of n doubles (one workgroup: n <= 1152^2, a few microseconds)
__global__ __launch_bounds__(1024) void k_sumsq(const double* __restrict__ x, size_t n, double* out) {
    __shared__ double sred[16];
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sred[w];
        out[0] = t;
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
hipError_t fill64(hipStream_t s, double* d_p, size_t n, double v) {
    hipLaunchKernelGGL(k_fill64, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, d_p, n, v);
    return hipGetLastError();
}

// d_A (n x n column-major, lower triangle read) -> d_L = its Cholesky factor (lower, zeros above the diagonal), d_Linv =
// L^-1 (likewise), d_scal[0] = trace(A^-1) = ||L^-1||_F^2; *d_status |= 2 if A is not positive definite.  d_A is left
// untouched; d_tmp: n x 32 doubles of scratch; d_minus: 32 doubles (filled here with -1).
size_t potrf_tmp_elems(int n) { return (size_t)2 * kChB * ((size_t)n + kChB) + 2 * kChB; }

hipError_t potrf_inverse(hipStream_t s, int n, const double* d_A, double* d_L, double* d_Linv, double* d_tmp, double* d_scal,
                         int* d_status) {
    if (n < 1) return hipErrorInvalidValue;
    double* d_diag = d_tmp + (size_t)kChB * n;                  // one 32 x 32 block per panel
    double* d_minus = d_diag + (size_t)kChB * (n + kChB);       // 32 x (-1)
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
            e = gemm64s(s, below, below, nb, l21, 1, n, l21, n, 1, a22, 1, n, nullptr, d_minus, nullptr, a22, 1, n);
            if (e != hipSuccess) return e;
        }
    }
    hipLaunchKernelGGL(k_potrf_finish, dim3(256), dim3(256), 0, s, d_L, n, d_diag);
    // L^-1 by block rows: X[i, 0:i) = -X_ii (L[i, 0:i) X[0:i, 0:i))
    for (int j0 = kChB; j0 < n; j0 += kChB) {
        const int nb = std::min(kChB, n - j0);
        e = gemm64s(s, nb, j0, j0, d_L + j0, 1, n, d_Linv, 1, n, d_tmp, 1, nb);
        if (e != hipSuccess) return e;
        e = gemm64s(s, nb, j0, nb, d_Linv + (size_t)j0 * n + j0, 1, n, d_tmp, 1, nb, d_Linv + j0, 1, n, nullptr, d_minus);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(1024), 0, s, d_Linv, (size_t)n * n, d_scal);
    return hipGetLastError();
}

}  // namespace nlek
