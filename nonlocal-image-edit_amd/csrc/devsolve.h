// Device dense solvers as the train pipeline uses them (devsolve.hip over dense64.hip).  Internal to libnle_hip.so.
#pragma once
#include "pipeline_internal.h"

namespace nlep {

// p x p problems go to the device from this order on (NLE_DEV_SOLVER_MIN overrides; NLE_HOST_SOLVER=1 keeps them on the
// host).  Below it a host core wins: the reduction is a chain of n dependent steps and a step costs the device two
// hand-offs between workgroups (measured: profiles/r3_dense_solver_timing.txt).
int dev_solver_min_n();
bool use_dev_solver(int n);

// holds `n` compute units of the DEVICE's budget for persistent launches while alive
int device_cu_count(int device);
struct CuLease {
    int dev, n;
    CuLease(int device, int n);
    ~CuLease();
    CuLease(const CuLease&) = delete;
    CuLease& operator=(const CuLease&) = delete;
};

// Symmetric eigen-computation of a device matrix in the reference's conventions (lower triangle read, eigenvalues
// descending, src/filter.cpp:207-212).
struct DevSymEig {
    int n = 0;
    hipStream_t st = nullptr;
    DevBuf<double> pub, tde;
    DevBuf<int> status;
    std::vector<double> d, e, D, hZ;  // T's diagonal and sub-diagonal (e[0] = 0), all eigenvalues DESCENDING; staging
    // workspace for order n; everything below runs on `stream` (null: the ctx stream).  Called by reduce() if need be;
    // call it earlier when the workspace has to be taken from the ctx cache at a particular moment (a second stream)
    void prepare(nle_ctx* c, int n, hipStream_t stream = nullptr);
    // reduction + eigenvalues; synchronises the stream and fills d, e, D.  false: the device cannot run the persistent
    // reduction (fewer compute units than workgroups), a hand-off timed out, or an eigenvalue is not finite -- the caller
    // falls back to the host solver (and, with more than one rank, agrees on that with its peers first)
    bool reduce(nle_ctx* c, int n, const double* d_M, const double* d_diag_add);
    // eigenvectors of D[first .. first + count) into d_Z (n x count column-major, device), enqueued on the stream
    // (hZ is the upload's staging buffer: it must outlive the copy, i.e. this object must)
    void vectors(nle_ctx* c, int first, int count, double* d_Z);
};

// Cholesky factor and its inverse on the device
struct DevChol {
    int n = 0;
    hipStream_t st = nullptr;
    DevBuf<double> L, Linv, tmp;
    DevBuf<int> status;
    double inv_trace = 0.0;
    bool ok = false;
    void prepare(nle_ctx* c, int n, hipStream_t stream = nullptr);
    void factor(nle_ctx* c, int n, const double* d_A);  // enqueue
    bool finish(nle_ctx* c);                            // synchronise: positive definite? inv_trace = trace(A^-1)
};

// the ctx's second stream (created on first use): the root of Wa runs there beside the Gram kernels
hipStream_t aux_stream(nle_ctx* c);

}  // namespace nlep
