#!/usr/bin/env python3
"""Wall times of the device dense solvers against the host's, by matrix order (Q-like spectrum; K eigenvectors)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    nle = entry.load_package()
    ctx = nle.Context(0)
    for n, K in [(int(a), int(b)) for a, b in (x.split(":") for x in os.environ.get("SIZES", "200:50,300:50,400:50,600:100,800:100,900:100,1152:100").split(","))]:
        rng = np.random.default_rng(n)
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.concatenate([[0.99999855], np.linspace(0.92, 1e-3, n - 1) ** 2])
        M = (Q * lam) @ Q.T
        M = 0.5 * (M + M.T)
        out = {"n": n, "K": K}
        for g in (os.environ.get("GS", "0").split(",")):
            if g != "0":
                if (-(-n // int(g)) + 3) * ((n + 1) & ~1) * 8 + 64 > 150 * 1024:
                    continue
                os.environ["NLE_SYTRD_G"] = g
            for name, fn in (("dev_values_ms", lambda: ctx.sym_eigen_device(M, 0, 0)),
                             ("dev_topK_ms", lambda: ctx.sym_eigen_device(M, 0, K)),
                             ("dev_chol_ms", lambda: ctx.cholesky_device(M + np.eye(n)))):
                fn()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(1e3 * (time.perf_counter() - t0))
                out[name + ("" if g == "0" else "_G" + g)] = round(min(ts), 3)
        os.environ.pop("NLE_SYTRD_G", None)
        for name, fn in (("host_topK_ms", lambda: nle.eigen_decomposition_top(M, K)),):
            fn()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                ts.append(1e3 * (time.perf_counter() - t0))
            out[name] = round(min(ts), 3)
        # the Cholesky entry point returns 2 n^2 doubles: where do its milliseconds go?  The same call into buffers allocated once
        # (pageable) and into page-locked ones (nle_host_alloc)
        Mp = M + np.eye(n)
        for name, bufs in (("dev_chol_prealloc_pageable_ms", (np.zeros(n * n), np.zeros(n * n))),
                           ("dev_chol_pinned_ms", (ctx.host_alloc((n * n,), dtype=np.float64), ctx.host_alloc((n * n,), dtype=np.float64)))):
            ctx.cholesky_device(Mp, out=bufs)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                ctx.cholesky_device(Mp, out=bufs)
                ts.append(1e3 * (time.perf_counter() - t0))
            out[name] = round(min(ts), 3)
        out["note"] = "device times include the n^2 upload and the result downloads"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
