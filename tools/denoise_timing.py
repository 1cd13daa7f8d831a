#!/usr/bin/env python3
"""Timing of the denoise wrapper's pieces at 4096 x 4096: bilateral prefilter, train on the
filtered plane, two chroma applies, colour conversions."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    cfg = synth.CONFIGS["cfg4"]
    H, W = cfg["H"], cfg["W"]
    ctx = nle.Context(0)
    base = synth.synthetic_luminance(H, W)
    rr, cc = np.mgrid[0:H, 0:W]
    bgr = np.stack([np.clip(base * 0.8 + 20 * np.sin(cc / 97.0), 0, 255), base,
                    np.clip(base * 0.9 + 25 * np.cos(rr / 61.0), 0, 255)], axis=-1).astype(np.uint8)
    d_bgr = torch.as_tensor(bgr, device="cuda:0")

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps, out

    res = {}
    res["bgr2lab_ms"], (lab, L) = timed(lambda: ctx.bgr2lab8(d_bgr))
    for sc, ss in ((10, 10), (10, 3)):
        res[f"bilateral_sigma{sc}_{ss}_ms"], Y = timed(lambda: ctx.bilateral8(L, sc, ss))
    Y = ctx.bilateral8(L, 10, 10)
    f = nle.NLEFilter(ctx)
    res["train_ms"], _ = timed(lambda: f.train_filter(Y, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"]))
    t = np.minimum(f.eigvals, 1.0) ** 2.0
    a = ctx.lab8_channel(lab, 1)
    res["apply_one_plane_ms"], ya = timed(lambda: f.apply(a, t))
    res["lab2bgr_planes_ms"], _ = timed(lambda: ctx.lab2bgr8(lab, Y, ya.reshape(H, W), ya.reshape(H, W)))
    res["taps_per_pixel_sigma_space_10"] = int((np.hypot(*np.mgrid[-15:16, -15:16]) <= 15).sum())
    print(json.dumps({"workload": f"{H}x{W}", **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}}))


if __name__ == "__main__":
    main()
