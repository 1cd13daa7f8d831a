#!/usr/bin/env python3
"""Per-kernel-class HIP-event times (nle_ctx_profile level 2: every launch) of train + apply_layers on a config."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    cfg = synth.CONFIGS[name]
    H, W, L = cfg["H"], cfg["W"], cfg["L"]
    ctx = nle.Context(0)
    if len(sys.argv) > 2:
        ctx.set_mode(int(sys.argv[2]))
    lum = torch.from_numpy(np.ascontiguousarray(synth.synthetic_luminance(H, W), dtype=np.float32)).cuda()
    steps = 3
    for it in range(steps + 1):
        if it == 1:
            ctx.profile(2)
        f = nle.NLEFilter(ctx)
        f.train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
        f.apply_layers(lum, L)
        torch.cuda.synchronize()
    st = ctx.kernel_stats()
    out = {k: {"launches_per_step": v[0] / steps, "avg_us": round(1e3 * v[1] / v[0], 2), "ms_per_step": round(v[1] / steps, 3)}
           for k, v in st.items() if v[0]}
    print(json.dumps({"config": name, "kernels": out, "gpu_ms_per_step": round(sum(v[1] for v in st.values()) / steps, 3)}))


if __name__ == "__main__":
    main()
