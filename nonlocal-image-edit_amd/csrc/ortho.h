// The orthogonalisation stage (reference src/filter.cpp:282-331) and what it shares with the train orchestration: the
// Nystrom factors of K_A, the host / device forms of `orthogonalize`.  Implemented in ortho.hip, used by pipeline.hip.
#pragma once
#include "devsolve.h"

namespace nlep {

// every eigenvalue of an SPD matrix with trace(M^-1) <= kCholMaxInvTrace is >= 1e-9 > NLE_EPS
constexpr double kCholMaxInvTrace = 1e9;

// Ka and its Cholesky factors left on the device by solve_Ka's device route (then VA and B stay empty on the host)
struct KaDevice {
    DevBuf<double> Ka;
    DevChol ch;
};

struct Nystrom {
    int r = 0, ldr = 0;
    bool chol = false;
    std::shared_ptr<KaDevice> dev;  // set: L = dev->ch.L, L^-1 = dev->ch.Linv (p x p column-major, device)
    std::vector<double> VA;   // p x r col-major
    std::vector<double> lam;  // r
    std::vector<double> B;    // p x r col-major
    std::vector<double> Ka;   // p x p (kept for the W blocks)
};

struct Ortho {
    int q = 0, K = 0, r_wa = 0, r_q = 0;
    std::vector<double> Sq, Cproj, VArows, Wa;
};

// ---- orthogonalisation in sample space (Phi-free path) ----
// Inputs: V_A (p x r), lambda, the two final Sinkhorn scaling vectors and
//   Gk = sum over NON-sample pixels of c_i^2 k_i k_i^T  (p x p, from k_gram_fused).
// With P = V_r V_r^T (projector on range(Ka); I when r == p) and Kr = V_r L V_r^T:
//   phi_a L phi_j^T = (P k_j)[a]  for a sample a and a pixel j, so (reference :247-250, q = r)
//   Wa  = R_A Kr[:q,:q] C_A,   Wab Wab^T = R_A (P Gk' P)[:q,:q] R_A,
//   Gk' = Gk + sum_{samples a >= q} c_a^2 Kr[:,a] Kr[:,a]^T   (samples that fall in the B block),
//   V_j = c_j k_j^T D,  D = P[:, :q] R_A T2,  T2 = S Vq Sq^-1/2   (:327), V_A rows = Wa T2.
// No 1/lambda factor appears anywhere: the ill-conditioned B = V_A / lambda is only used for
// the r-vectors of the Sinkhorn update.
struct OrthoSS {
    int q = 0, K = 0, r_wa = 0, r_q = 0;
    bool chol_wa = false;
    std::vector<double> Sq, D, Vrows;  // D: p x K, Vrows: p x K (col-major)
    // state between the two halves
    int p = 0, r = 0;
    std::vector<double> cA, rA, Kr, P, Wa, S, St, A2;  // Q = A2 + S^T (Wab Wab^T) S,  St = S^T
};

// reference :282-331 on the host from the materialised quantities (G = sum over all pixels of c^2 phi phi^T)
Ortho orthogonalize_host(const Nystrom& ny, int p, const std::vector<double>& u_c, const std::vector<double>& u_r,
                         std::vector<double> G, int n_eig, bool device_f32 = true, int topk_solver = 0);
// sample-space form, host: everything that does not need the Gram matrix, then the rest
void ortho_ss_prepare(OrthoSS& o, const Nystrom& ny, int p, const std::vector<double>& sA_c, const std::vector<double>& sA_r,
                      bool literal_q = false);  // literal_q: S = Wa^-1/2 itself (no Cholesky root): Q is then the reference's matrix
void ortho_ss_finish(OrthoSS& o, std::vector<double> Gk, int n_eig, int topk_solver = 0);
// sample-space form with the p x p products (and, from dev_solver_min_n() samples on, the solvers) on the device
void ortho_ss_device(nle_ctx* c, OrthoSS& o, const Nystrom& ny, int p, const std::vector<double>& sA_c,
                     const std::vector<double>& sA_r, double* d_Gk, int n_eig, const std::function<void()>& enqueue_gram,
                     const std::function<void()>& reduce_gram, double* host_ms, double* host_overlapped_ms, Trace& tr);

}  // namespace nlep
