"""Regenerates tests/golden/readme_oracle.json: what the CPU oracle (oracle/nle_oracle.py) computes on every README
sample pair (tests/readme_pairs.py) -- the eigenvalue counts kept by the three 1e-10 cuts (src/filter.cpp:214 applied
to Ka :262, Wa :287, Q :313), the eigenvalues either side of each cut, K', the kept eigenvalues of Q, the per-layer L2
norms, and the distance of the oracle's 8-bit L plane from the author's output image.

    python tests/golden/make_readme_golden.py
"""
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import readme_pairs as rp  # noqa: E402


def load_bgr(path):
    return np.asarray(Image.open(path).convert("RGB"))[..., ::-1].copy()  # BGR like cv::imread


def run_pair(oracle, pair):
    """everything the tests compare, for one pair (shared by the CPU and GPU tests)"""
    name, _, _, nr, nc, hx, hy, T, K, w = pair
    src_p, want_p = rp.paths(pair)
    src, want = load_bgr(src_p), load_bgr(want_p)
    lab = oracle.bgr_to_lab8(src)
    L = lab[..., 0].astype(np.float64)
    info = []
    V, S = oracle.train_filter(L, nr, nc, hx, hy, T, K, info=info)
    wts = [float(x) for x in w]
    layers = oracle.apply_layers(V, S, L, len(wts)).reshape(len(wts), -1)
    y = oracle.apply_filter(V, L, oracle.transform_eigenvalues(S, wts))
    L_out = np.rint(np.clip(y, 0, 255)).astype(np.uint8)                     # src/filter.cpp:434-436
    lab2 = lab.copy()
    lab2[..., 0] = L_out
    out = oracle.lab8_to_bgr(lab2)
    L_want = oracle.bgr_to_lab8(want)[..., 0].astype(np.float64)
    err = np.abs(oracle.bgr_to_lab8(out)[..., 0].astype(np.float64) - L_want)
    return dict(name=name, src=src, want=want, L=L, S=S, layers=layers, L_out=L_out, out=out, info=info,
                mean=float(err.mean()), p99=float(np.percentile(err, 99)), moved=float(np.abs(L_want - L).mean()))


def main():
    oracle = entry.load_oracle()
    rec = {}
    for pair in rp.PAIRS:
        r = run_pair(oracle, pair)
        rec[r["name"]] = dict(shape=list(r["L"].shape), args=rp.cli_args(pair), cuts=r["info"],
                              K_out=int(r["S"].size), eigvals=[float(x) for x in r["S"]],
                              layer_norms=[float(np.linalg.norm(l)) for l in r["layers"]],
                              vs_readme_output=dict(moved=r["moved"], mean=r["mean"], p99=r["p99"]))
        print(r["name"], rec[r["name"]]["cuts"], rec[r["name"]]["vs_readme_output"], flush=True)
    with open(os.path.join(HERE, "readme_oracle.json"), "w") as fh:
        json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
