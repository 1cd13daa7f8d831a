#!/usr/bin/env python3
"""Per-layer relative L2 of the device path vs the CPU oracle, both formulations (GPU box).
    python tools/parity_report.py [--big]      (--big adds cfg2 512x512 and a 1024x1024 p=200 case)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

CASES = [
    (48, 64, 4, 5, 16.0, 30.0, 10, 8, 4),
    (96, 128, 6, 8, 32.0, 30.0, 10, 10, 4),
    (15, 20, 10, 7, 8.0, 30.0, 5, 6, 3),
    (33, 47, 3, 4, 20.0, 25.0, 1, 4, 2),
    (64, 64, 8, 8, 16.0, 30.0, 7, 70, 5),
    (267, 400, 10, 20, 100.0, 30.0, 50, 30, 4),
]
BIG = [(512, 512, 10, 20, 128.0, 30.0, 10, 10, 4), (1024, 1024, 20, 10, 256.0, 30.0, 10, 50, 4)]


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    nle, oracle = entry.load_package(), entry.load_oracle()
    ctx = nle.Context(0)
    cases = CASES + (BIG if "--big" in sys.argv else [])
    for (H, W, nr, nc, hx, hy, T, K, L) in cases:
        x = oracle.synthetic_luminance(H, W)
        if H * W <= 300000:
            V_o, S_o, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
            Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
            lam_min = inter["lam"][-1]
        else:
            V_o, S_o = oracle.train_filter_streaming(x, nr, nc, hx, hy, T, K)
            Y_o = oracle.apply_layers_streaming(V_o, S_o, x, L)
            lam_min = float("nan")
        for m, name in ((1, "materialised"), (2, "phi_free"), (3, "phi_free_exp")):
            ctx.set_mode(m)
            f = nle.NLEFilter(ctx).train_filter(x.astype(np.float32), nr, nc, hx, hy, T, K)
            Y = f.apply_layers(x.astype(np.float32), L).cpu().numpy().astype(np.float64)
            errs = " ".join("%.1e" % rel(Y[j], Y_o[j]) for j in range(L))
            print(f"{H}x{W} p={f.info()['p']} r={f.info()['r']} K={f.info()['K']} T={T} lam_min={lam_min:.1e} "
                  f"{name:13s} eigvals {rel(f.eigvals, S_o):.1e}  layers {errs}", flush=True)
            f.close()
    ctx.close()


if __name__ == "__main__":
    main()
