#!/usr/bin/env python3
"""Replay an input saved by tools/parity_fuzz.py under every formulation and both factorisations of K_A.
    python tools/parity_replay.py gpurun_out/fuzz_fail_*.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def main():
    nle, oracle = entry.load_package(), entry.load_oracle()
    ctx = nle.Context(0)
    for path in sys.argv[1:]:
        d = np.load(path)
        x = d["x"]
        nr, nc, hx, hy, T, K, L, _ = d["params"]
        nr, nc, T, K, L = int(nr), int(nc), int(T), int(K), int(L)
        V_o, S_o, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
        Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
        print(os.path.basename(path), x.shape, (nr, nc, hx, hy, T, K, L), "lam_min %.2e" % inter["lam"][-1], "S_o[:3]", S_o[:3],
              "layer norms", [float(np.linalg.norm(Y_o[j])) for j in range(L)])
        for mode in (0, 1, 2, 3):
            for force in ("", "1"):
                if force:
                    os.environ["NLE_FORCE_EIG"] = "1"
                else:
                    os.environ.pop("NLE_FORCE_EIG", None)
                ctx.set_mode(mode)
                try:
                    f = nle.NLEFilter(ctx).train_filter(x.astype(np.float32), nr, nc, hx, hy, T, K)
                    Y = f.apply_layers(x.astype(np.float32), L).cpu().numpy().astype(np.float64)
                    print("  mode", mode, "force_eig" if force else "default  ", "eig rel %.2e" % rel(f.eigvals, S_o),
                          "eig abs", np.abs(np.array(f.eigvals) - S_o)[:3], "layers", ["%.1e" % rel(Y[j], Y_o[j]) for j in range(L)])
                    f.close()
                finally:
                    ctx.set_mode(0)
        os.environ.pop("NLE_FORCE_EIG", None)


if __name__ == "__main__":
    main()
