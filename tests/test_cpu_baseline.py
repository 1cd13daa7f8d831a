"""oracle/nle_cpu_baseline.cpp -- the C++17 + OpenMP streaming restatement bench.py times as `cpu_baseline` -- against the numpy
oracle (oracle/nle_oracle.py, itself pinned by the reference's test cases and README pairs): same ranks at the 1e-10 cuts,
eigenvalues and layer norms to rounding.  Host only."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "oracle", "nle_cpu_baseline.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpub") / "nle_cpu_baseline"
    subprocess.run(["g++", "-O2", "-mavx2", "-mfma", "-fopenmp", "-std=c++17", SRC, "-o", str(out)], check=True)
    return str(out)


@pytest.mark.parametrize("case", [(96, 128, 6, 8, 32.0, 30.0, 10, 10, 4), (48, 64, 4, 5, 16.0, 30.0, 10, 8, 4),
                                  (61, 83, 7, 9, 700.0, 30.0, 6, 12, 3)])   # the last: hx so wide that the cuts truncate
def test_cpp_baseline_matches_the_numpy_oracle(oracle, exe, case):
    H, W, nr, nc, hx, hy, T, K, L = case
    r = subprocess.run([exe] + [str(v) for v in (H, W, nr, nc, hx, hy, T, K, L, 2)], capture_output=True, text=True, timeout=300,
                       check=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    x = oracle.synthetic_luminance(H, W)
    inter = {}
    V, S = oracle.train_filter_streaming(x, nr, nc, hx, hy, T, K, intermediates=inter)
    Y = oracle.apply_layers_streaming(V, S, x, L)
    assert d["r"] == inter["lam"].size and d["r_wa"] == inter["cuts"][0]["kept"] and d["r_q"] == inter["cuts"][1]["kept"]
    assert d["K"] == S.size
    assert np.abs(np.array(d["eigvals"]) - S).max() < 1e-9
    norms = np.linalg.norm(Y, axis=1)
    assert np.abs(np.array(d["layer_norms"]) / norms - 1).max() < 1e-7
    pr = np.array(d["probes"]).reshape(L, 8)
    idx = [(H * W - 1) * j // 7 for j in range(8)]
    assert np.abs(pr - Y[:, idx]).max() <= 1e-5 * np.abs(Y).max()
