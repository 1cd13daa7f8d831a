import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
import torch
nle = entry.load_package()
synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
cfg = dict(cfg); cfg["T"] = 2
lum = torch.as_tensor(synth.synthetic_luminance(cfg["H"], cfg["W"]).astype(np.float32), device="cuda:0")
ctx = nle.Context(0)
f = nle.NLEFilter(ctx).train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
torch.cuda.synchronize()
