#!/usr/bin/env python3
"""Host (p-sized) time of a train with the default full eigensolve of Q and with the opt-in Lanczos top-K solver
(reference src/filter.cpp:170-199): ms per image and the `host` share reported by
nle_filter_timings, at cfg4 and cfg5.   python tools/host_algebra_timing.py [cfg ...]"""
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
    for name in (sys.argv[1:] or ["cfg4", "cfg5"]):
        cfg = synth.CONFIGS[name]
        H, W, L = cfg["H"], cfg["W"], cfg["L"]
        lum = torch.from_numpy(np.ascontiguousarray(synth.synthetic_luminance(H, W), dtype=np.float32)).cuda()
        rec = {"config": name}
        evs = {}
        for solver, tag in ((0, "full"), (1, "lanczos")):
            ctx = nle.Context(0)
            ctx.set_topk_solver(solver)
            f = nle.NLEFilter(ctx)
            ts, hosts = [], []
            for it in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                f.train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                hosts.append(f.timings()["host"])
            evs[tag] = f.eigvals.copy()
            rec[tag] = {"train_ms_median": round(float(np.median(ts[1:])) * 1e3, 3), "host_ms_median": round(float(np.median(hosts[1:])), 3),
                        "K": int(f.info()["K"])}
            f.close()
            ctx.close()
        k = min(evs["full"].size, evs["lanczos"].size)
        rec["eigenvalue_rel_diff"] = float(np.linalg.norm(evs["full"][:k] - evs["lanczos"][:k]) / np.linalg.norm(evs["full"][:k]))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
