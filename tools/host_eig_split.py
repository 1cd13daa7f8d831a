#!/usr/bin/env python3
"""Where the host's top-K eigensolve of Q spends its time (n = 196, K = 50 as at cfg4): all-host solver with K = 1 and
K = 50 columns, and the device-reduction variant (nle_eigen_decomposition_top_device)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
nle = entry.load_package()
rng = np.random.default_rng(1)
n = 196
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
lam = np.sort(np.concatenate([1.0 - 0.5 * rng.random(60), 10.0 ** (-10 * rng.random(n - 60))]))[::-1]
M = (Q * lam) @ Q.T
M = 0.5 * (M + M.T)
ctx = nle.Context(0)
def best(f, reps=30):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print("host, K = 1 : %.3f ms" % best(lambda: nle.eigen_decomposition_top(M, 1)))
print("host, K = 50: %.3f ms" % best(lambda: nle.eigen_decomposition_top(M, 50)))
print("device reduction + host rest, K = 50: %.3f ms" % best(lambda: ctx.eigen_decomposition_top_device(M, 50)))
print("device reduction + host rest, K = 1 : %.3f ms" % best(lambda: ctx.eigen_decomposition_top_device(M, 1)))
