import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as entry
nle = entry.load_package()
ctx = nle.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 900
rng = np.random.default_rng(1)
X = rng.standard_normal((n, n + 5))
M = X @ X.T / n + 1e-3 * np.eye(n)
for _ in range(3):
    ctx.cholesky_device(M)
