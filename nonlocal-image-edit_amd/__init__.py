"""Python mirror of the C ABI in include/nle.h (ctypes; no compute happens in Python).

The product is `lib/libnle_hip.so` (HIP kernels for gfx950 + host pipeline).  This module
only binds it for tests and `bench.py`, using torch tensors as device memory and the
torch current stream as the HIP stream -- plumbing, not the product.  There is no CPU
fallback: every compute call needs the shared library and a HIP device and raises
otherwise.  Nothing here imports `oracle/`.

Names follow the reference (`include/filter.hpp:20-54`): `NLEFilter.train_filter`
<-> `NLEFilter::trainFilter`, `.apply` <-> `::apply`, `transform_eigenvalues`
<-> `transformEigenValues`, `eigen_decomposition` <-> `eigenDecomposition`, ...
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnle_hip.so")

# the signature table and the integer constants are generated from include/nle.h (tools/gen_ctypes.py -> _abi.py;
# tests/test_abi.py regenerates and compares), so this mirror cannot drift from the header
from . import _abi
from ._abi import (ALLREDUCE_FN, NLE_ERR_COMM, NLE_ERR_HIP, NLE_ERR_INVALID, NLE_ERR_NUMERIC, NLE_OK)  # noqa: F401

EPS = 1e-10  # NLE_EPS
KERNEL_COUNT = _abi.NLE_KERNEL_COUNT
_P = C.c_void_p
_SIGNATURES = _abi.SIGNATURES
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
MODE_AUTO, MODE_MATERIALISED, MODE_PHI_FREE, MODE_PHI_FREE_EXP, MODE_MATERIALISED_F64, MODE_STREAMED_F64 = (
    _abi.NLE_MODE_AUTO, _abi.NLE_MODE_MATERIALISED, _abi.NLE_MODE_PHI_FREE, _abi.NLE_MODE_PHI_FREE_EXP,
    _abi.NLE_MODE_MATERIALISED_F64, _abi.NLE_MODE_STREAMED_F64)

_lib = None


class NLEError(RuntimeError):
    """Non-zero status from the C ABI (the C++ surface maps these to std::runtime_error)."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def lib() -> C.CDLL:
    """Load libnle_hip.so (built by build.py).  Fails loudly if it is missing."""
    global _lib
    if _lib is None:
        # NLE_LIB_PATH: a measurement build of the same sources (tools/abl_run.sh) instead of the product library -- so that
        # such a build never has to be copied over lib/libnle_hip.so
        path = os.environ.get("NLE_LIB_PATH") or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: run `python nonlocal-image-edit_amd/build.py` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        # torch-rocm ships its own copy of the HIP runtime: load torch first so that this process ends up
        # with ONE libamdhip64 (loading ours first leaves torch and the library on different runtimes and
        # hipGetDeviceCount then reports no device)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _check(status: int, ctx=None):
    if status != NLE_OK:
        msg = lib().nle_last_error(ctx)
        raise NLEError(status, (msg or b"").decode() or f"nle status {status}")


# ------------------------------------------------------------------ host-only helpers
def ld(n: int) -> int:
    return int(lib().nle_ld(int(n)))


def rccl_unique_id() -> bytes:
    buf = (C.c_char * 128)()
    _check(lib().nle_rccl_unique_id(buf, 128))
    return bytes(buf)


def lab8_tables():
    """tables of the fixed-point 8-bit BGR -> Lab conversion (nle_lab8_tables): gamma[256], cbrt[3072], coeffs[3][3]"""
    g = np.zeros(256, dtype=np.uint16)
    c = np.zeros(3072, dtype=np.uint16)
    k = np.zeros(9, dtype=np.int32)
    _check(lib().nle_lab8_tables(_np_ptr(g), _np_ptr(c), _np_ptr(k)))
    return g, c, k.reshape(3, 3)


def lab8_inverse_tables():
    """tables of the integer 8-bit Lab -> BGR conversion (nle_lab8_inverse_tables): yf[256][2], ab_to_xz[36864] (index
    t + 8145), inv_gamma[4096], coeffs[3][3] (rows R, G, B)"""
    yf = np.zeros(512, dtype=np.uint16)
    ab = np.zeros(36864, dtype=np.int32)
    ig = np.zeros(4096, dtype=np.uint16)
    k = np.zeros(9, dtype=np.int32)
    _check(lib().nle_lab8_inverse_tables(_np_ptr(yf), _np_ptr(ab), _np_ptr(ig), _np_ptr(k)))
    return yf.reshape(256, 2), ab, ig, k.reshape(3, 3)


def sample_grid(H, W, n_row_samples, n_col_samples):
    """`samplePixels` (src/filter.cpp:56-80) in closed form.
    Returns dict(row_step,row_off,n_sel_rows,col_step,col_off,n_sel_cols)."""
    out = [C.c_int() for _ in range(6)]
    st = lib().nle_sample_grid(H, W, n_row_samples, n_col_samples, *[C.byref(o) for o in out])
    if st != NLE_OK:
        raise NLEError(st, "Number of samples per row and col must be <= that of image.")
    keys = ("row_step", "row_off", "n_sel_rows", "col_step", "col_off", "n_sel_cols")
    return dict(zip(keys, (o.value for o in out)))


def slab_rows(H, rank, world):
    r0, r1 = C.c_int(), C.c_int()
    st = lib().nle_slab_rows(H, rank, world, C.byref(r0), C.byref(r1))
    if st != NLE_OK:
        raise NLEError(st, "bad slab arguments")
    return r0.value, r1.value


def eigen_decomposition(M: np.ndarray, eps: float = EPS):
    """`eigenDecomposition` (src/filter.cpp:204-228): returns (U, D), descending, cut at eps."""
    M = np.asfortranarray(np.asarray(M, dtype=np.float64))
    n = M.shape[0]
    if M.ndim != 2 or M.shape[1] != n or n == 0:
        raise NLEError(NLE_ERR_INVALID, "eigen_decomposition needs a non-empty square matrix")
    U = np.zeros((n, n), dtype=np.float64, order="F")
    D = np.zeros(n, dtype=np.float64)
    r = C.c_int()
    st = lib().nle_eigen_decomposition(_np_ptr(M), n, float(eps), _np_ptr(U), _np_ptr(D), C.byref(r))
    if st != NLE_OK:
        raise NLEError(st, "eigensolver did not converge")
    return np.ascontiguousarray(U[:, :r.value]), D[:r.value].copy()


def eigen_decomposition_top(M: np.ndarray, kmax: int, eps: float = EPS):
    """all eigenvalues (descending) and the first min(kmax, n) eigenvectors: (U [n x min(kmax, n)], D [n], r)"""
    M = np.asfortranarray(np.asarray(M, dtype=np.float64))
    n = M.shape[0]
    k = min(int(kmax), n)
    U = np.zeros((n, k), dtype=np.float64, order="F")
    D = np.zeros(n, dtype=np.float64)
    r = C.c_int()
    st = lib().nle_eigen_decomposition_top(_np_ptr(M), n, float(eps), int(kmax), _np_ptr(U), _np_ptr(D), C.byref(r))
    if st != NLE_OK:
        raise NLEError(st, "eigensolver did not converge")
    return np.ascontiguousarray(U), D, r.value


def eigen_decomposition_topk(M: np.ndarray, kmax: int, eps: float = EPS):
    """the min(kmax, n) largest eigenvalues (descending), their eigenvectors and the number of eigenvalues >= eps, as the
    train path computes them for Q on the host (bisection when 2 kmax <= n): (U [n x k], Dk [k], r)"""
    M = np.asfortranarray(np.asarray(M, dtype=np.float64))
    n = M.shape[0]
    k = min(int(kmax), n)
    U = np.zeros((n, k), dtype=np.float64, order="F")
    D = np.zeros(k, dtype=np.float64)
    r = C.c_int()
    st = lib().nle_eigen_decomposition_topk(_np_ptr(M), n, float(eps), int(kmax), _np_ptr(U), _np_ptr(D), C.byref(r))
    if st != NLE_OK:
        raise NLEError(st, "eigensolver did not converge")
    return np.ascontiguousarray(U), D, r.value


def topk_eigen_decomposition(M: np.ndarray, n_largest: int, eps: float = EPS):
    """`topkEigenDecomposition` (src/filter.cpp:170-199, the USE_SPECTRA build): Lanczos top-k; returns (U, D)."""
    M = np.asfortranarray(np.asarray(M, dtype=np.float64))
    n = M.shape[0]
    nev = min(int(n_largest), n - 1)
    U = np.zeros((n, nev), dtype=np.float64, order="F")
    D = np.zeros(nev, dtype=np.float64)
    r = C.c_int()
    st = lib().nle_topk_eigen_decomposition(_np_ptr(M), n, int(n_largest), float(eps), _np_ptr(U), _np_ptr(D), C.byref(r))
    if st != NLE_OK:
        raise NLEError(st, "Lanczos did not converge")
    return np.ascontiguousarray(U[:, :r.value]), D[:r.value].copy()


def transform_eigenvalues(eigvals, weights):
    """`transformEigenValues` (src/filter.cpp:334-347)."""
    ev = np.ascontiguousarray(eigvals, dtype=np.float64)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    out = np.zeros_like(ev)
    st = lib().nle_transform_eigenvalues(_np_ptr(ev), ev.size, _np_ptr(w), w.size, _np_ptr(out))
    if st != NLE_OK:
        raise NLEError(st, "bad arguments")
    return out


def layer_responses(eigvals, n_layers):
    ev = np.ascontiguousarray(eigvals, dtype=np.float64)
    out = np.zeros((n_layers, ev.size), dtype=np.float64)
    st = lib().nle_layer_responses(_np_ptr(ev), ev.size, n_layers, _np_ptr(out))
    if st != NLE_OK:
        raise NLEError(st, "bad arguments")
    return out


# ------------------------------------------------------------------ device side
def _torch():
    import torch
    return torch


class Context:
    """`nle_ctx`: one per process/GPU.  Uses torch's current stream on `device`."""

    def __init__(self, device: int = 0, rank: int = 0, world: int = 1, allreduce=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device: nonlocal-image-edit_amd has no CPU fallback")
        self.device = int(device)
        torch.cuda.set_device(self.device)
        self._h = C.c_void_p()
        self._pinned = []
        # A dedicated torch stream: its handle is never the null stream (a NULL stream argument means
        # "create your own" in the C ABI), torch collectives issued under it are ordered with the
        # kernels of the ctx, and `_sync_in` orders the ctx after whatever produced its inputs.
        self._stream = torch.cuda.Stream(device=self.device)
        st = lib().nle_ctx_create(self.device, C.c_void_p(self._stream.cuda_stream), C.byref(self._h))
        if st != NLE_OK:
            raise NLEError(st, (lib().nle_last_error(None) or b"").decode())
        self.rank, self.world = rank, world
        self.slab_input = False
        self._cb = None
        self._comm = None
        self._allreduce = allreduce

    def set_shard(self, rank: int, world: int, n_samples: int, allreduce):
        """`allreduce(tensor)` sums a float64 CUDA tensor in place over all ranks (e.g.
        torch.distributed.all_reduce)."""
        torch = _torch()
        n = int(lib().nle_comm_len(int(n_samples)))
        self._comm = torch.zeros(n, dtype=torch.float64, device=f"cuda:{self.device}")
        comm = self._comm
        base = comm.data_ptr()

        debug_path = os.environ.get("NLE_DEBUG_COMM")
        views = {}   # (offset, count) -> tensor view: the Sinkhorn loop calls this 2T times with the same slice
        stream_ctx = torch.cuda.stream

        def _cb(user, d_buf, count):
            try:
                key = (int(d_buf), int(count))
                view = views.get(key)
                if view is None:
                    off = (key[0] - base) // 8
                    view = views[key] = comm[off:off + key[1]]
                if debug_path:
                    with open(debug_path + f".rank{rank}", "a") as fh:
                        fh.write(f"{key[1]}\n")
                with stream_ctx(self._stream):   # same stream as the ctx's kernels
                    allreduce(view)
                return 0
            except Exception as e:  # noqa: BLE001 - must not propagate through C
                print("nle allreduce callback failed:", repr(e), flush=True)
                return 1

        self._cb = ALLREDUCE_FN(_cb)
        _check(lib().nle_ctx_set_shard(self._h, rank, world, self._cb, None, C.c_void_p(base), n), self._h)
        self.rank, self.world = rank, world

    def init_rccl(self, rank: int, world: int, unique_id: bytes):
        """native RCCL all-reduces (nle_ctx_init_rccl): collective over the `world` ranks; `unique_id` = the 128 bytes of
        rccl_unique_id() made by rank 0"""
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        _check(lib().nle_ctx_init_rccl(self._h, int(rank), int(world), buf, 128), self._h)
        self.rank, self.world = rank, world

    def abort_rccl(self):
        """nle_ctx_abort_rccl: the error path of a multi-rank host (collectives of this ctx fail from here on)"""
        _check(lib().nle_ctx_abort_rccl(self._h))

    def synchronize(self):
        _check(lib().nle_ctx_synchronize(self._h), self._h)

    def set_mode(self, mode: int):
        """0 auto, 1 materialised Phi, 2 Phi-free (NLE_MODE_* in include/nle.h)."""
        _check(lib().nle_ctx_set_mode(self._h, int(mode)), self._h)

    def set_nystrom_bf16x3(self, on: bool = True):
        """the fused Nystrom GEMM on the bf16 matrix cores with split operands (nle_ctx_set_nystrom_bf16x3)"""
        _check(lib().nle_ctx_set_nystrom_bf16x3(self._h, 1 if on else 0), self._h)

    def host_alloc(self, shape, dtype=np.float32):
        """page-locked host array (nle_host_alloc).  The block belongs to the ctx and is freed by Context.close(): the
        returned array (and every view of it) must not be used after that"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        _check(lib().nle_host_alloc(self._h, n, C.byref(ptr)), self._h)
        buf = (C.c_char * max(n, 1)).from_address(ptr.value)
        self._pinned.append(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def eigen_decomposition_top_device(self, M, kmax: int, eps: float = EPS):
        """eigen_decomposition_top with the tridiagonal reduction on the GPU (nle_eigen_decomposition_top_device; n <= 224)"""
        M = np.asfortranarray(np.asarray(M, dtype=np.float64))
        n = M.shape[0]
        k = min(int(kmax), n)
        U = np.zeros((n, k), dtype=np.float64, order="F")
        D = np.zeros(n, dtype=np.float64)
        r = C.c_int()
        _check(lib().nle_eigen_decomposition_top_device(self._h, _np_ptr(M), n, float(eps), int(kmax), _np_ptr(U), _np_ptr(D),
                                                        C.byref(r)), self._h)
        return np.ascontiguousarray(U), D, r.value

    def sym_eigen_device(self, M, first: int = 0, count: int = 0, eps: float = EPS):
        """all eigenvalues (descending) and the eigenvectors of D[first:first+count] with the reduction, the bisection and the
        back-transformation on the GPU (nle_sym_eigen_device; 3 <= n <= 1152).  Returns (U n x count, D, r)."""
        M = np.asfortranarray(np.asarray(M, dtype=np.float64))
        n = M.shape[0]
        U = np.zeros((n, max(count, 1)), dtype=np.float64, order="F")
        D = np.zeros(n, dtype=np.float64)
        r = C.c_int()
        _check(lib().nle_sym_eigen_device(self._h, _np_ptr(M), n, float(eps), int(first), int(count), _np_ptr(U), _np_ptr(D),
                                          C.byref(r)), self._h)
        return np.ascontiguousarray(U[:, :count]), D, r.value

    def cholesky_device(self, M, out=None):
        """(L, L^-1, trace(M^-1), ok) of a symmetric matrix (lower triangle read) on the GPU (nle_cholesky_device).
        out = (L, Linv): caller's n x n float64 buffers (flat or 2-D, e.g. page-locked from host_alloc), filled
        COLUMN-major and returned as they are"""
        M = np.asfortranarray(np.asarray(M, dtype=np.float64))
        n = M.shape[0]
        if out is None:
            L = np.zeros((n, n), dtype=np.float64, order="F")
            Li = np.zeros((n, n), dtype=np.float64, order="F")
        else:
            L, Li = out
            if any(a.dtype != np.float64 or a.size != n * n for a in (L, Li)):
                raise NLEError(NLE_ERR_INVALID, "cholesky_device: out buffers must hold n * n float64 values")
        tr = C.c_double()
        ok = C.c_int()
        _check(lib().nle_cholesky_device(self._h, _np_ptr(M), n, _np_ptr(L), _np_ptr(Li), C.byref(tr), C.byref(ok)), self._h)
        if out is not None:
            return L, Li, tr.value, bool(ok.value)
        return np.ascontiguousarray(L), np.ascontiguousarray(Li), tr.value, bool(ok.value)

    def set_slab_input(self, on: bool = True):
        """planes passed to train / apply hold this rank's rows only (nle_ctx_set_slab_input); pass shape=(H, W)"""
        _check(lib().nle_ctx_set_slab_input(self._h, 1 if on else 0), self._h)
        self.slab_input = bool(on)

    def set_topk_solver(self, solver: int):
        """0: full eigensolve of Q (the reference's default build); 1: Lanczos top-K (its USE_SPECTRA build)"""
        _check(lib().nle_ctx_set_topk_solver(self._h, int(solver)), self._h)

    def trim(self):
        """release the cached device workspace"""
        _check(lib().nle_ctx_trim(self._h), self._h)

    def profile(self, enable=True):
        """Per-kernel HIP-event timing on the ctx's stream (resets the counters).  True / 2: every
        kernel; 1: the N-sized kernels only (cheaper: each timed launch costs ~10 us of stream gaps)."""
        level = 2 if enable is True else int(enable)
        _check(lib().nle_ctx_profile(self._h, level), self._h)

    def kernel_stats(self):
        """{kernel name: (launches, total_ms)} accumulated since `profile(True)`."""
        out = {}
        for kid in range(KERNEL_COUNT):
            n, ms = C.c_longlong(), C.c_double()
            _check(lib().nle_ctx_kernel_stats(self._h, kid, C.byref(n), C.byref(ms)), self._h)
            out[lib().nle_kernel_name(kid).decode()] = (n.value, ms.value)
        return out

    def close(self):
        if self._h:
            for ptr in self._pinned:
                lib().nle_host_free(self._h, ptr)
            self._pinned = []
            lib().nle_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _sync_in(self):
        """order the ctx stream after the work already queued on torch's current stream"""
        torch = _torch()
        self._stream.wait_stream(torch.cuda.current_stream(self.device))

    # ---- stage-level entry points (device tensors in/out) ----
    def _lum(self, lum):
        torch = _torch()
        t = torch.as_tensor(lum, dtype=torch.float32, device=f"cuda:{self.device}").contiguous()
        if t.ndim != 2:
            raise NLEError(NLE_ERR_INVALID, "luminance must be H x W")
        self._sync_in()
        return t

    def local_pixels(self, H, W):
        r0, r1 = slab_rows(H, self.rank, self.world)
        return (r1 - r0) * W

    def compute_kernel(self, lum, n_row_samples, n_col_samples, hx, hy, want_kab=True):
        """`computeKernel` (src/filter.cpp:114-167): returns (Ka [p x p fp64 numpy],
        kab [n_local x ld(p) fp32 CUDA tensor, natural pixel order])."""
        torch = _torch()
        lum = self._lum(lum)
        H, W = lum.shape
        if n_row_samples > H or n_col_samples > W:
            raise NLEError(NLE_ERR_INVALID, "Number of samples per row and col must be <= that of image.")
        g = sample_grid(H, W, n_row_samples, n_col_samples)
        p = g["n_sel_rows"] * g["n_sel_cols"]
        Ka = np.zeros((p, p), dtype=np.float64, order="F")
        kab = None
        if want_kab:
            kab = torch.empty((self.local_pixels(H, W), ld(p)), dtype=torch.float32, device=lum.device)
        _check(lib().nle_compute_kernel(self._h, C.c_void_p(lum.data_ptr()), H, W, n_row_samples, n_col_samples,
                                        float(hx), float(hy), _np_ptr(Ka),
                                        C.c_void_p(kab.data_ptr()) if want_kab else None), self._h)
        return np.ascontiguousarray(Ka), kab

    def nystrom(self, lum, n_row_samples, n_col_samples, hx, hy):
        """`nystromApproximation` fused with the affinity evaluation: (eigvals, phi[n_local x ld(r)])."""
        torch = _torch()
        lum = self._lum(lum)
        H, W = lum.shape
        if n_row_samples > H or n_col_samples > W:
            raise NLEError(NLE_ERR_INVALID, "Number of samples per row and col must be <= that of image.")
        g = sample_grid(H, W, n_row_samples, n_col_samples)
        p = g["n_sel_rows"] * g["n_sel_cols"]
        n_local = self.local_pixels(H, W)
        buf = torch.empty(n_local * ld(p), dtype=torch.float32, device=lum.device)
        ev = np.zeros(p, dtype=np.float64)
        r = C.c_int()
        _check(lib().nle_nystrom(self._h, C.c_void_p(lum.data_ptr()), H, W, n_row_samples, n_col_samples, float(hx),
                                 float(hy), _np_ptr(ev), C.byref(r), C.c_void_p(buf.data_ptr())), self._h)
        rr = r.value
        return ev[:rr].copy(), buf[: n_local * ld(rr)].view(n_local, ld(rr)), rr

    def ts_gemm(self, A, kd, B):
        """C = A[:, :kd] @ B  (A fp32 CUDA M x lda, B fp64 numpy kd x nc) -> CUDA M x ld(nc)."""
        torch = _torch()
        self._sync_in()
        B = np.asfortranarray(np.asarray(B, dtype=np.float64))
        M, lda = A.shape
        nc = B.shape[1]
        Cc = torch.empty((M, ld(nc)), dtype=torch.float32, device=A.device)
        _check(lib().nle_ts_gemm(self._h, C.c_void_p(A.data_ptr()), M, lda, kd, _np_ptr(B), nc,
                                 C.c_void_p(Cc.data_ptr())), self._h)
        return Cc

    def sinkhorn_scalings(self, phi, r, eigvals, max_iter=10):
        """Sinkhorn iterations (src/filter.cpp:238-245) on device phi -> (u_c, u_r)."""
        ev = np.ascontiguousarray(eigvals, dtype=np.float64)
        uc, ur = np.zeros(r), np.zeros(r)
        self._sync_in()
        M, ldp = phi.shape
        _check(lib().nle_sinkhorn_scalings(self._h, C.c_void_p(phi.data_ptr()), M, ldp, r, _np_ptr(ev), max_iter,
                                           _np_ptr(uc), _np_ptr(ur)), self._h)
        return uc, ur

    def gram(self, phi, r, u):
        self._sync_in()
        u = np.ascontiguousarray(u, dtype=np.float64)
        G = np.zeros((r, r), dtype=np.float64, order="F")
        M, ldp = phi.shape
        _check(lib().nle_gram(self._h, C.c_void_p(phi.data_ptr()), M, ldp, r, _np_ptr(u), _np_ptr(G)), self._h)
        return np.ascontiguousarray(G)

    def row_scalings(self, phi, r, u):
        torch = _torch()
        self._sync_in()
        u = np.ascontiguousarray(u, dtype=np.float64)
        M, ldp = phi.shape
        out = torch.empty(M, dtype=torch.float64, device=phi.device)
        _check(lib().nle_row_scalings(self._h, C.c_void_p(phi.data_ptr()), M, ldp, r, _np_ptr(u),
                                      C.c_void_p(out.data_ptr())), self._h)
        return out

    # ---- colour / denoise wrapper pieces (device tensors) ----
    def bgr2lab8(self, bgr):
        """`cvtColor(COLOR_BGR2Lab)` on an H x W x 3 uint8 image: (lab uint8 H x W x 3, L float32 H x W)"""
        torch = _torch()
        t = torch.as_tensor(bgr, dtype=torch.uint8, device=f"cuda:{self.device}").contiguous()
        self._sync_in()
        H, W = t.shape[:2]
        lab = torch.empty_like(t)
        L = torch.empty((H, W), dtype=torch.float32, device=t.device)
        _check(lib().nle_bgr2lab8(self._h, C.c_void_p(t.data_ptr()), H * W, C.c_void_p(lab.data_ptr()),
                                  C.c_void_p(L.data_ptr())), self._h)
        return lab, L

    def lab8_channel(self, lab, channel):
        torch = _torch()
        self._sync_in()
        H, W = lab.shape[:2]
        out = torch.empty((H, W), dtype=torch.float32, device=lab.device)
        _check(lib().nle_lab8_channel(self._h, C.c_void_p(lab.data_ptr()), H * W, int(channel),
                                      C.c_void_p(out.data_ptr())), self._h)
        return out

    def lab2bgr8(self, lab, L=None, a=None, b=None):
        """`max/min/convertTo(CV_8U)/merge/cvtColor(COLOR_Lab2BGR)` with optional float32 replacement planes"""
        torch = _torch()
        self._sync_in()
        H, W = lab.shape[:2]
        out = torch.empty_like(lab)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _check(lib().nle_lab2bgr8_planes(self._h, C.c_void_p(lab.data_ptr()), ptr(L), ptr(a), ptr(b), H * W,
                                         C.c_void_p(out.data_ptr())), self._h)
        return out

    def bilateral8(self, plane, sigma_color, sigma_space):
        """`cv::bilateralFilter(plane, -1, sigmaColor, sigmaSpace)` on an integer-valued float32 plane"""
        torch = _torch()
        src = self._lum(plane)
        H, W = src.shape
        out = torch.empty_like(src)
        _check(lib().nle_bilateral8(self._h, C.c_void_p(src.data_ptr()), H, W, float(sigma_color), float(sigma_space),
                                    C.c_void_p(out.data_ptr())), self._h)
        return out

    def bench_affinity(self, lum, n_row_samples, n_col_samples, hx, hy, reps=10):
        torch = _torch()
        lum = self._lum(lum)
        H, W = lum.shape
        g = sample_grid(H, W, n_row_samples, n_col_samples)
        p = g["n_sel_rows"] * g["n_sel_cols"]
        kab = torch.empty((self.local_pixels(H, W), ld(p)), dtype=torch.float32, device=lum.device)
        ms = C.c_double()
        _check(lib().nle_bench_affinity(self._h, C.c_void_p(lum.data_ptr()), H, W, n_row_samples, n_col_samples,
                                        float(hx), float(hy), C.c_void_p(kab.data_ptr()), reps, C.byref(ms)), self._h)
        return ms.value, kab

    def bench_affinity64(self, lum, n_row_samples, n_col_samples, hx, hy, rows, reps=10):
        """average launch time (ms) of the fp64 affinity kernel on the first `rows` image rows; returns (ms, bytes written)"""
        torch = _torch()
        lum = self._lum(lum)
        H, W = lum.shape
        g = sample_grid(H, W, n_row_samples, n_col_samples)
        p = g["n_sel_rows"] * g["n_sel_cols"]
        rows = int(min(rows, H))
        kab = torch.empty((rows * W, ld(p)), dtype=torch.float64, device=lum.device)
        ms = C.c_double()
        _check(lib().nle_bench_affinity64(self._h, C.c_void_p(lum.data_ptr()), H, W, n_row_samples, n_col_samples,
                                          float(hx), float(hy), rows, C.c_void_p(kab.data_ptr()), reps, C.byref(ms)), self._h)
        return ms.value, kab.numel() * 8

    def bench_sinkhorn_pass(self, phi, r, reps=10):
        self._sync_in()
        ms = C.c_double()
        M, ldp = phi.shape
        _check(lib().nle_bench_sinkhorn_pass(self._h, C.c_void_p(phi.data_ptr()), M, ldp, r, reps, C.byref(ms)),
               self._h)
        return ms.value


class NLEFilter:
    """`nle::NLEFilter` (include/filter.hpp:35-54) over the C ABI; state = V, eigvals on the GPU."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._f = C.c_void_p()
        self.shape = None

    def close(self):
        if self._f:
            lib().nle_filter_destroy(self._f)
            self._f = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _full_shape(self, plane, shape):
        """(H, W) of the full image for a plane handed in: the plane's own shape, or -- Context.set_slab_input -- the
        `shape` given, the plane then being this rank's rows [row0, row1) (nle_slab_rows)"""
        if not getattr(self.ctx, "slab_input", False) or self.ctx.world == 1:
            return tuple(int(v) for v in plane.shape)
        if shape is None:
            raise NLEError(NLE_ERR_INVALID, "slab input: pass shape=(H, W) of the full image")
        H, W = int(shape[0]), int(shape[1])
        r0, r1 = slab_rows(H, self.ctx.rank, self.ctx.world)
        if tuple(plane.shape) != (r1 - r0, W):
            raise NLEError(NLE_ERR_INVALID, f"slab input: rank {self.ctx.rank} of {self.ctx.world} must pass rows "
                                            f"[{r0}, {r1}) x {W}, got {tuple(plane.shape)}")
        return H, W

    def train_filter(self, lum, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter=10, n_eigen_vectors=5, shape=None):
        """`NLEFilter::trainFilter` (src/filter.cpp:480-502); lum: H x W luminance (CUDA tensor
        stays on the device; numpy is uploaded).  Slab-input contexts pass their rows and shape=(H, W)."""
        self.close()
        lum = self.ctx._lum(lum)
        H, W = self._full_shape(lum, shape)
        _check(lib().nle_train(self.ctx._h, C.c_void_p(lum.data_ptr()), H, W, int(n_row_samples), int(n_col_samples),
                               float(hx), float(hy), int(n_sinkhorn_iter), int(n_eigen_vectors), C.byref(self._f)),
               self.ctx._h)
        self.shape = (H, W)
        return self

    def train_filter_host(self, lum, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter=10, n_eigen_vectors=5,
                          shape=None):
        """nle_train_host: the H x W fp32 plane is a HOST array (pinned: Context.host_alloc); the filter keeps the
        uploaded plane for apply_layers_host(None, ...).  Slab-input contexts pass their rows and shape=(H, W)."""
        self.close()
        lum = np.ascontiguousarray(lum, dtype=np.float32)
        H, W = self._full_shape(lum, shape)
        _check(lib().nle_train_host(self.ctx._h, _np_ptr(lum), H, W, int(n_row_samples), int(n_col_samples), float(hx),
                                    float(hy), int(n_sinkhorn_iter), int(n_eigen_vectors), C.byref(self._f)), self.ctx._h)
        self.shape = (H, W)
        return self

    def train_filter_host_u8(self, lum8, n_row_samples, n_col_samples, hx, hy, n_sinkhorn_iter=10, n_eigen_vectors=5,
                             shape=None):
        """nle_train_host_u8: the plane as the 8-bit levels themselves (a HOST uint8 array): a quarter of the upload; the
        same filter as train_filter_host on those levels as floats"""
        self.close()
        lum8 = np.ascontiguousarray(lum8, dtype=np.uint8)
        H, W = self._full_shape(lum8, shape)
        _check(lib().nle_train_host_u8(self.ctx._h, _np_ptr(lum8), H, W, int(n_row_samples), int(n_col_samples), float(hx),
                                       float(hy), int(n_sinkhorn_iter), int(n_eigen_vectors), C.byref(self._f)), self.ctx._h)
        self.shape = (H, W)
        return self

    def apply_layers_host(self, x, n_layers, out):
        """nle_apply_layers_host: x a HOST H x W fp32 array or None (= the training plane kept by
        train_filter_host); out: HOST (L, n_local) fp32 array"""
        H, W = self.shape
        n_local = self.info()["n_local"]
        xp = None
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            want = n_local if (self.ctx.slab_input and self.ctx.world > 1) else H * W
            if x.size != want:
                raise NLEError(NLE_ERR_INVALID, f"apply_layers_host: x has {x.size} values, the filter needs {want}")
            xp = _np_ptr(x)
        # the library writes n_layers * n_local floats into `out`: a short array would be a host heap overflow
        if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags.c_contiguous
                and out.size == int(n_layers) * n_local):
            raise NLEError(NLE_ERR_INVALID, f"apply_layers_host: out must be a C-contiguous float32 array of "
                                            f"{int(n_layers)} x {n_local} values")
        _check(lib().nle_apply_layers_host(self._f, xp, H, W, int(n_layers), _np_ptr(out)), self.ctx._h)
        return out

    def level_tiles(self):
        """(first, count) of the 16-level tiles the table kernels of this filter work on (nle_filter_level_tiles)"""
        a, b = C.c_int(), C.c_int()
        _check(lib().nle_filter_level_tiles(self._f, C.byref(a), C.byref(b)), self.ctx._h)
        return a.value, b.value

    def apply_u8_host(self, x, f_s, out):
        """nle_apply_u8_host: the 8-bit L plane `enhance` merges back (src/filter.cpp:428-436) -- apply, clamp to
        [0, 255], round half to even.  x a HOST H x W fp32 array or None (= the training plane); out: HOST uint8 array of
        n_local values"""
        H, W = self.shape
        n_local = self.info()["n_local"]
        fs = np.ascontiguousarray(f_s, dtype=np.float64)
        if fs.ndim != 1 or fs.size != self.info()["K"]:
            raise NLEError(NLE_ERR_INVALID, f"f_s must hold K' = {self.info()['K']} values, got {fs.shape}")
        xp = None
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            want = n_local if (self.ctx.slab_input and self.ctx.world > 1) else H * W
            if x.size != want:
                raise NLEError(NLE_ERR_INVALID, f"apply_u8_host: x has {x.size} values, the filter needs {want}")
            xp = _np_ptr(x)
        if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous and out.size == n_local):
            raise NLEError(NLE_ERR_INVALID, f"apply_u8_host: out must be a C-contiguous uint8 array of {n_local} values")
        _check(lib().nle_apply_u8_host(self._f, xp, H, W, _np_ptr(fs), _np_ptr(out)), self.ctx._h)
        return out

    def apply_rounded8(self, x, f_s, out=None):
        """nle_apply_rounded8: the clamped, rounded plane as fp32 levels (the replacement-channel argument of lab2bgr8)"""
        torch = _torch()
        x = self.ctx._lum(x)
        H, W = self._full_shape(x, self.shape)
        fs = np.ascontiguousarray(f_s, dtype=np.float64)
        if fs.ndim != 1 or fs.size != self.info()["K"]:
            raise NLEError(NLE_ERR_INVALID, f"f_s must hold K' = {self.info()['K']} values, got {fs.shape}")
        n = self.info()["n_local"]
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=x.device)
        _check(lib().nle_apply_rounded8(self._f, C.c_void_p(x.data_ptr()), H, W, _np_ptr(fs), C.c_void_p(out.data_ptr())),
               self.ctx._h)
        return out

    def apply_u8(self, x, f_s, out=None):
        """nle_apply_u8: the same on a device-resident plane; returns a uint8 device tensor of n_local values"""
        torch = _torch()
        x = self.ctx._lum(x)
        H, W = self._full_shape(x, self.shape)
        fs = np.ascontiguousarray(f_s, dtype=np.float64)
        if fs.ndim != 1 or fs.size != self.info()["K"]:
            raise NLEError(NLE_ERR_INVALID, f"f_s must hold K' = {self.info()['K']} values, got {fs.shape}")
        n = self.info()["n_local"]
        if out is None:
            out = torch.empty(n, dtype=torch.uint8, device=x.device)
        _check(lib().nle_apply_u8(self._f, C.c_void_p(x.data_ptr()), H, W, _np_ptr(fs), C.c_void_p(out.data_ptr())),
               self.ctx._h)
        return out

    def info(self):
        n = C.c_longlong()
        v = [C.c_int() for _ in range(5)]
        _check(lib().nle_filter_info(self._f, C.byref(n), *[C.byref(x) for x in v]))
        return dict(n_local=n.value, K=v[0].value, r=v[1].value, p=v[2].value, row0=v[3].value, row1=v[4].value)

    def diag(self):
        """what the last train decided (nle_filter_diag): formulation taken, ranks kept by the three 1e-10 cuts"""
        v = np.zeros(8, dtype=np.int32)
        _check(lib().nle_filter_diag(self._f, _np_ptr(v)))
        keys = ("formulation", "p", "r_Ka", "r_Wa", "r_Q", "K", "chol_Ka", "chol_Wa")
        return {k: int(x) for k, x in zip(keys, v)}

    @property
    def eigvals(self):
        K = self.info()["K"]
        out = np.zeros(K, dtype=np.float64)
        _check(lib().nle_filter_eigvals(self._f, _np_ptr(out)))
        return out

    def eigvecs(self):
        """m_eigvecs as a CUDA tensor view (n_local x ld(K)), pixel order."""
        torch = _torch()
        ptr, ldv = C.c_void_p(), C.c_int()
        _check(lib().nle_filter_eigvecs(self._f, C.byref(ptr), C.byref(ldv)))
        n = self.info()["n_local"]
        out = torch.empty((n, ldv.value), dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        _check(lib().nle_filter_copy_eigvecs(self._f, C.c_void_p(out.data_ptr())), self.ctx._h)
        return out

    def eigvec_range(self, ncols):
        """(min, max) of the first `ncols` eigenvector columns over this rank's pixels (the banner of
        src/filter.cpp:506); does not materialise an implicit V"""
        mn = np.zeros(ncols, dtype=np.float64)
        mx = np.zeros(ncols, dtype=np.float64)
        _check(lib().nle_filter_eigvec_range(self._f, int(ncols), _np_ptr(mn), _np_ptr(mx)), self.ctx._h)
        return mn, mx

    def timings(self):
        ms = np.zeros(6)
        _check(lib().nle_filter_timings(self._f, _np_ptr(ms)))
        return dict(zip(("setup", "sinkhorn", "gram", "project", "host", "total"), ms.tolist()))

    def apply(self, x, f_s, out=None):
        """`NLEFilter::apply` (src/filter.cpp:445-458): y = V diag(fS) V^T x (local slab)."""
        torch = _torch()
        x = self.ctx._lum(x)
        H, W = self._full_shape(x, self.shape)
        fs = np.ascontiguousarray(f_s, dtype=np.float64)
        if fs.ndim != 1 or fs.size != self.info()["K"]:   # nle_apply reads K doubles
            raise NLEError(NLE_ERR_INVALID, f"f_s must hold K' = {self.info()['K']} values, got {fs.shape}")
        n = self.info()["n_local"]
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=x.device)
        _check(lib().nle_apply(self._f, C.c_void_p(x.data_ptr()), H, W, _np_ptr(fs), C.c_void_p(out.data_ptr())),
               self.ctx._h)
        return out

    def apply_layers(self, x, n_layers, out=None):
        torch = _torch()
        x = self.ctx._lum(x)
        H, W = self._full_shape(x, self.shape)
        n = self.info()["n_local"]
        if out is None:
            out = torch.empty((n_layers, n), dtype=torch.float32, device=x.device)
        _check(lib().nle_apply_layers(self._f, C.c_void_p(x.data_ptr()), H, W, int(n_layers),
                                      C.c_void_p(out.data_ptr())), self.ctx._h)
        return out
