#!/usr/bin/env python3
"""How the table kernels depend on the image's level distribution (LDS same-address atomics).

The histogram kernels accumulate per-level sums with LDS atomics, so their speed depends on how many lanes of a
wave share a level: none for a noisy image, all of them for a flat region.  This times cfg4's train+apply on
several 4096^2 level distributions and prints the per-launch time of the table kernels, one JSON line each:
  synthetic   the bench image
  uniform     independent uniform levels (no structure, conflict-free rate)
  flat_rows   every image row one level (all 64 lanes of every wave on one address)
  blocks      64x64 flat blocks with 5% noisy pixels (flat regions with texture)
  checker     two levels alternating per pixel (two hot addresses per wave)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def images(H, W, synth):
    rng = np.random.default_rng(5)
    yield "synthetic", synth.synthetic_luminance(H, W)
    yield "uniform", rng.integers(0, 256, (H, W)).astype(np.float32)
    yield "flat_rows", np.repeat(rng.integers(0, 256, (H, 1)), W, axis=1).astype(np.float32)
    blocks = np.kron(rng.integers(0, 256, (H // 64, W // 64)), np.ones((64, 64), dtype=np.int64))
    noise = rng.random((H, W)) < 0.05
    blocks = np.where(noise, rng.integers(0, 256, (H, W)), blocks)
    yield "blocks", blocks.astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    yield "checker", np.where((yy + xx) % 2 == 0, 60, 190).astype(np.float32)


def main():
    import torch
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
    H, W, L = cfg["H"], cfg["W"], cfg["L"]
    ctx = nle.Context(0)
    for name, img in images(H, W, synth):
        lum = torch.from_numpy(np.ascontiguousarray(img)).cuda()
        try:
            for it in range(3):
                if it == 1:
                    ctx.profile(1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                f = nle.NLEFilter(ctx)
                f.train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
                f.apply_layers(lum, L)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 2
            st = ctx.kernel_stats()
            ctx.profile(0)
            per = {k: round(v[1] / v[0], 4) for k, v in st.items() if v[0]}
            print(json.dumps({"image": name, "ms_per_step": round(ms, 2), "ms_per_launch": per}), flush=True)
        except Exception as e:  # noqa: BLE001  (e.g. an image whose K_A has fewer than K eigenpairs)
            ctx.profile(0)
            print(json.dumps({"image": name, "error": str(e)[:200]}), flush=True)


if __name__ == "__main__":
    main()
