import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
nle = entry.load_package()
synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
H, W, L = cfg["H"], cfg["W"], cfg["L"]
ctx = nle.Context(0)
lum = torch.from_numpy(np.ascontiguousarray(synth.synthetic_luminance(H, W), dtype=np.float32)).cuda()
f = nle.NLEFilter(ctx)
import time
for it in range(5):
    if it == 4: os.environ["NLE_TRACE"] = "1"
    t0 = time.perf_counter()
    f.train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
    t1 = time.perf_counter()
    Y = f.apply_layers(lum, L)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"[step {it}] train {1e3 * (t1 - t0):.2f} ms, apply {1e3 * (t2 - t1):.2f} ms", file=sys.stderr)
