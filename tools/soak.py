#!/usr/bin/env python3
"""Soak: many train + apply steps on one ctx; per-step time must stay flat and device memory must not grow
(the ctx arena caches workspace between calls).  python tools/soak.py [steps] [config]"""
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    name = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    cfg = synth.CONFIGS[name]
    H, W, L = cfg["H"], cfg["W"], cfg["L"]
    ctx = nle.Context(0)
    lum = torch.from_numpy(np.ascontiguousarray(synth.synthetic_luminance(H, W), dtype=np.float32)).cuda()
    out = torch.empty((L, H * W), dtype=torch.float32, device="cuda")
    f = nle.NLEFilter(ctx)
    times, free = [], []
    ev0 = None
    for it in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f.train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
        f.apply_layers(lum, L, out=out)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        if it % 50 == 0 or it == steps - 1:
            free.append(torch.cuda.mem_get_info()[0])
        ev = np.array(f.eigvals)
        if ev0 is None:
            ev0 = ev
        assert np.max(np.abs(ev - ev0)) < 1e-9, "eigenvalues drifted"
    t = np.array(times[5:])
    print(json.dumps({"config": name, "steps": steps, "ms_median": float(np.median(t)), "ms_p99": float(np.percentile(t, 99)),
                      "ms_max": float(t.max()), "first_quarter_median": float(np.median(t[:len(t) // 4])),
                      "last_quarter_median": float(np.median(t[-len(t) // 4:])),
                      "free_bytes_samples": free, "free_bytes_drop_after_warmup": int(free[1] - free[-1]) if len(free) > 2 else 0}))


if __name__ == "__main__":
    main()
