"""Generates the committed golden vectors from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

The reference itself cannot be built here (needs Eigen3 + OpenCV), so these
vectors come from the oracle AFTER it passed the reference's own test cases
(tests/test_oracle_reference_cases.py) and the README flower pair (tests/test_oracle_flower.py).
Contents of small_cases.npz, for each case id:
    <id>/args   H W nRow nCol hx hy T K L
    <id>/x      the luminance plane (seeded synthetic, integer valued)
    <id>/lam    eigenvalues of Ka (descending, cut)          src/filter.cpp:262-271
    <id>/S      eigenvalues kept by orthogonalize            :313-316
    <id>/Y      per-layer outputs (L x H*W)                  :334-347, :456
    <id>/rc     Sinkhorn scalings r, c at 8 probe pixels     :238-245
flower_cfg1.npz: config 1 (flower-50.bmp with the README args): lam, S, per-layer L2 norms,
the 8-bit L plane in and out."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

CASES = {
    "c48x64": (48, 64, 4, 5, 16.0, 30.0, 10, 8, 4),
    "c96x128": (96, 128, 6, 8, 32.0, 30.0, 10, 10, 4),
    "c15x20_step1": (15, 20, 10, 7, 8.0, 30.0, 5, 6, 3),
    "c33x47_T1": (33, 47, 3, 4, 20.0, 25.0, 1, 4, 2),
}


def main():
    oracle = entry.load_oracle()
    out = {}
    for cid, (H, W, nr, nc, hx, hy, T, K, L) in CASES.items():
        x = oracle.synthetic_luminance(H, W)
        V, S, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
        Y = oracle.apply_layers(V, S, x, L).reshape(L, -1)
        probe = np.linspace(0, H * W - 1, 8).astype(np.int64)
        nat_r = np.empty(H * W)
        nat_c = np.empty(H * W)
        nat_r[inter["perm"]] = inter["r"]
        nat_c[inter["perm"]] = inter["c"]
        out[f"{cid}/args"] = np.array([H, W, nr, nc, hx, hy, T, K, L], dtype=np.float64)
        out[f"{cid}/x"] = x.astype(np.uint8)
        out[f"{cid}/lam"] = inter["lam"]
        out[f"{cid}/S"] = S
        out[f"{cid}/Y"] = Y
        out[f"{cid}/rc"] = np.stack([probe.astype(np.float64), nat_r[probe], nat_c[probe]])
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **out)

    from PIL import Image
    bgr = np.asarray(Image.open(os.path.join(HERE, "flower-50.bmp")).convert("RGB"))[..., ::-1].copy()
    lab = oracle.bgr_to_lab8(bgr)
    Lp = lab[..., 0].astype(np.float64)
    V, S, inter = oracle.train_filter(Lp, 10, 20, 100.0, 30.0, 50, 30, return_intermediates=True)
    Y = oracle.apply_layers(V, S, Lp, 4).reshape(4, -1)
    y = oracle.apply_filter(V, Lp, oracle.transform_eigenvalues(S, [2.0, 3.0, 4.0, 1.0]))
    np.savez_compressed(os.path.join(HERE, "flower_cfg1.npz"), L_in=lab[..., 0], lam=inter["lam"], S=S,
                        layer_norms=np.linalg.norm(Y, axis=1),
                        L_out=np.rint(np.clip(y, 0, 255)).astype(np.uint8),
                        Y_probe=Y[:, ::997].copy())
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
