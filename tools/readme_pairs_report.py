"""Run the CPU oracle on every README sample pair (tests/readme_pairs.py) and print, per pair: p, the rank kept by
each of the three eigensolves with the eigenvalues either side of the 1e-10 cut, K', and how far the oracle's output is
from the author's output image (8-bit L plane).  `--sweep NAME` re-runs one pair with K_A's rank forced to r-8 .. r+8
to see which rank the author's Eigen build must have kept.  Test infrastructure (imports oracle/)."""
import argparse
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import readme_pairs as rp  # noqa: E402

oracle = entry.load_oracle()


def load_bgr(path):
    return np.asarray(Image.open(path).convert("RGB"))[..., ::-1].copy()


def stats(oracle_bgr, want_bgr, src_bgr):
    L_src = oracle.bgr_to_lab8(src_bgr)[..., 0].astype(np.float64)
    L_want = oracle.bgr_to_lab8(want_bgr)[..., 0].astype(np.float64)
    L_got = oracle.bgr_to_lab8(oracle_bgr)[..., 0].astype(np.float64)
    err = np.abs(L_got - L_want)
    return dict(moved=float(np.abs(L_want - L_src).mean()), mean=float(err.mean()), p99=float(np.percentile(err, 99)),
                max=float(err.max()), corr=float(np.corrcoef(L_got.ravel(), L_want.ravel())[0, 1]))


def run(pair, force_rank=None):
    name, _, _, nr, nc, hx, hy, T, K, w = pair
    src_p, want_p = rp.paths(pair)
    src, want = load_bgr(src_p), load_bgr(want_p)
    info = []
    t0 = time.time()
    out = oracle.enhance_image(src, nr, nc, hx, hy, T, K, [float(x) for x in w], info=info, force_rank=force_rank)
    rec = dict(name=name, shape=list(src.shape[:2]), args=rp.cli_args(pair), force_rank=force_rank,
               KA=info[0], WA=info[1], Q=info[2], K_out=min(K, info[2]["kept"]), seconds=round(time.time() - t0, 1))
    rec.update(stats(out, want, src))
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--sweep", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    recs = []
    if a.sweep:
        pair = [p for p in rp.PAIRS if p[0] == a.sweep][0]
        base = run(pair)
        print(json.dumps(base), flush=True)
        recs.append(base)
        r0 = base["KA"]["kept"]
        for r in range(max(1, r0 - 8), min(base["KA"]["n"], r0 + 8) + 1):
            if r == r0:
                continue
            rec = run(pair, force_rank=r)
            print(json.dumps(rec), flush=True)
            recs.append(rec)
    else:
        for pair in rp.PAIRS:
            if a.only and pair[0] != a.only:
                continue
            rec = run(pair)
            print(json.dumps(rec), flush=True)
            recs.append(rec)
    if a.out:
        with open(a.out, "w") as fh:
            for r in recs:
                fh.write(json.dumps(r) + "\n")
