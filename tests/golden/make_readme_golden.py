"""Regenerates tests/golden/readme_oracle.json: what the CPU oracle (oracle/nle_oracle.py) computes on every README
sample pair (tests/readme_pairs.py) -- the eigenvalue counts kept by the three 1e-10 cuts (src/filter.cpp:214 applied
to Ka :262, Wa :287, Q :313), the eigenvalues either side of each cut, K', the kept eigenvalues of Q, the per-layer L2
norms, and the distance of the oracle's 8-bit L plane from the author's output image.

    python tests/golden/make_readme_golden.py [pair ...]
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
    d_bgr = np.abs(out.astype(int) - want.astype(int))   # against the author's FILE, byte for byte
    return dict(name=name, src=src, want=want, L=L, S=S, layers=layers, L_out=L_out, out=out, info=info,
                mean=float(err.mean()), p99=float(np.percentile(err, 99)), moved=float(np.abs(L_want - L).mean()),
                bgr_exact=float((d_bgr == 0).mean()), bgr_max=int(d_bgr.max()))


# pairs whose oracle run is too long to repeat inside the GPU tests (rock2: 512k pixels, 600 samples, 2.5 min and ~12 GB on
# the CPU): what the tests compare is stored instead -- the L plane, the 8-bit output plane, the eigenvalues, the cuts,
# every layer's norm and every 61st pixel of every layer
STORED = ("rock2",)
PROBE_STEP = 61


def stored_path(name):
    return os.path.join(HERE, f"readme_{name}_oracle.npz")


def save_stored(r):
    np.savez_compressed(stored_path(r["name"]), L=r["L"].astype(np.uint8), L_out=r["L_out"], S=r["S"],
                        info=json.dumps(r["info"]), layer_norms=np.array([np.linalg.norm(l) for l in r["layers"]]),
                        layer_probes=r["layers"][:, ::PROBE_STEP].copy(), probe_step=PROBE_STEP,
                        mean=r["mean"], p99=r["p99"], moved=r["moved"], bgr_exact=r["bgr_exact"], bgr_max=r["bgr_max"])


def load_stored(name):
    z = np.load(stored_path(name))
    return dict(name=name, L=z["L"].astype(np.float64), L_out=z["L_out"], S=z["S"], info=json.loads(str(z["info"])),
                layer_norms=[float(v) for v in z["layer_norms"]], layer_probes=z["layer_probes"],
                probe_step=int(z["probe_step"]), mean=float(z["mean"]), p99=float(z["p99"]), moved=float(z["moved"]),
                bgr_exact=float(z["bgr_exact"]), bgr_max=int(z["bgr_max"]))


def main():
    oracle = entry.load_oracle()
    rec = {}
    only = sys.argv[1:]
    if only:  # regenerate some pairs, keep the others
        with open(os.path.join(HERE, "readme_oracle.json")) as fh:
            rec = json.load(fh)
    for pair in rp.PAIRS:
        if only and pair[0] not in only:
            continue
        r = run_pair(oracle, pair)
        if pair[0] in STORED:
            save_stored(r)
        rec[r["name"]] = dict(shape=list(r["L"].shape), args=rp.cli_args(pair), cuts=r["info"],
                              K_out=int(r["S"].size), eigvals=[float(x) for x in r["S"]],
                              layer_norms=[float(np.linalg.norm(l)) for l in r["layers"]],
                              vs_readme_output=dict(moved=r["moved"], mean=r["mean"], p99=r["p99"], bgr_exact=r["bgr_exact"],
                                                    bgr_max=r["bgr_max"]))
        print(r["name"], rec[r["name"]]["cuts"], rec[r["name"]]["vs_readme_output"], flush=True)
    with open(os.path.join(HERE, "readme_oracle.json"), "w") as fh:
        json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
