#!/usr/bin/env python3
"""HBM roofline of the materialising affinity pass (computeKernel's K_AB, reference src/filter.cpp:139-145):
B_A = N*4*(1+p) algorithmic bytes per launch / average launch time (HIP events,
nle_bench_affinity), for the BASELINE.json configs.   python tools/affinity_report.py [cfg ...]"""
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
    ctx = nle.Context(0)
    for name in (sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]):
        cfg = synth.CONFIGS[name]
        H, W = cfg["H"], cfg["W"]
        lum = torch.as_tensor(synth.synthetic_luminance(H, W).astype(np.float32), device="cuda:0")
        g = nle.sample_grid(H, W, cfg["n_row"], cfg["n_col"])
        p = g["n_sel_rows"] * g["n_sel_cols"]
        ms, kab = ctx.bench_affinity(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], reps=10)
        nbytes = H * W * 4.0 * (1 + p)
        gbs = nbytes / (ms * 1e-3) / 1e9
        print(json.dumps({"config": name, "kernel": "k_affinity", "N": H * W, "p": p, "avg_launch_ms": ms,
                          "algorithmic_bytes": nbytes, "achieved_GBs": gbs, "frac_of_8TBs": gbs / 8000.0,
                          "frac_of_measured_copy_6.29TBs": gbs / 6290.0}), flush=True)
        del kab, lum
    ctx.close()


if __name__ == "__main__":
    main()
