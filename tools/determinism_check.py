import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
import torch
nle = entry.load_package()
synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
H, W, L = cfg["H"], cfg["W"], cfg["L"]
ctx = nle.Context(0)
lum = torch.from_numpy(np.ascontiguousarray(synth.synthetic_luminance(H, W), dtype=np.float32)).cuda()
outs = []
for it in range(3):
    f = nle.NLEFilter(ctx)
    f.train_filter(lum, cfg["n_row"], cfg["n_col"], cfg["hx"], cfg["hy"], cfg["T"], cfg["K"])
    y = f.apply_layers(lum, L).cpu().numpy()
    outs.append((np.array(f.eigvals), y))
for i in (1, 2):
    de = np.max(np.abs(outs[i][0] - outs[0][0]) / np.abs(outs[0][0]))
    dy = np.linalg.norm(outs[i][1] - outs[0][1]) / np.linalg.norm(outs[0][1])
    print("run", i, "eigvals max rel diff", de, "layers rel L2 diff", dy, "bitwise equal outputs:", np.array_equal(outs[i][1], outs[0][1]))
