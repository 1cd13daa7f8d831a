"""Pins the CPU oracle where the reference actually operates: every sample pair of the reference's README table
(README.md:72-83: input image, enhance arguments, the author's output image; fixtures under tests/golden/).  Six of
the nine pairs are rank-truncated -- eigenvalues of Ka, Wa and Q sit right at the 1e-10 cut (src/filter.cpp:214 via
:262, :287, :313) -- so these are the inputs on which the truncation rules, `q = phi.cols()` (:247) and the
lower-triangle reading of the non-symmetric Wa (:287) matter.

The 8-bit Lab conversion on both sides of the path is OpenCV's in the reference (version unknown, fixed-point tables)
and a float restatement here, so the L plane the filter is trained on differs from the author's by rounding ties:
a LOOSE known-answer test, about one grey level.  `bird` is the one pair the restatement
misses by more (3.4 levels): tools/readme_pair_sensitivity.py shows that on this input a +-1 level change of 10 % of
the L pixels moves the output by 4.8 levels (flower: 0.16), while noise of 1e-13 on every affinity moves it by
nothing -- the miss is the unpinned Lab rounding amplified by an ill-conditioned example, not the hot path.
"""
import json
import os

import numpy as np
import pytest

import readme_pairs as rp
from conftest import GOLDEN

# mean / p99 of |L_oracle - L_author| in grey levels allowed per pair (measured: profiles/r2_readme_pairs_oracle.jsonl)
TOL = {name: (1.0, 8.0) for name in [p[0] for p in rp.PAIRS]}
TOL["bird"] = (4.0, 14.0)   # see the module docstring

_cache = {}


def oracle_run(oracle, name):
    if name not in _cache:
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_readme_golden", os.path.join(GOLDEN, "make_readme_golden.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        pair = [p for p in rp.PAIRS if p[0] == name][0]
        r = mod.run_pair(oracle, pair)
        _cache[name] = {k: r[k] for k in ("S", "layers", "info", "mean", "p99", "moved", "L_out", "L")}
        _cache[name]["layer_norms"] = [float(np.linalg.norm(l)) for l in r["layers"]]
    return _cache[name]


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "readme_oracle.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", [p[0] for p in rp.PAIRS])
def test_oracle_reproduces_readme_output(oracle, golden, name):
    r = oracle_run(oracle, name)
    g = golden[name]
    cuts = [(c["n"], c["kept"]) for c in r["info"]]
    print(f"{name}: p = {cuts[0][0]}, kept by the cuts Ka/Wa/Q = {[c[1] for c in cuts]}, K' = {r['S'].size}, "
          f"lambda at the cuts = {[(c['last_kept'], c['first_dropped']) for c in r['info']]}; filter moves L by "
          f"{r['moved']:.2f}; oracle vs author's output: mean |dL| = {r['mean']:.3f}, p99 = {r['p99']:.1f}")
    mean_tol, p99_tol = TOL[name]
    assert r["moved"] > 5.0                      # the edit is large ...
    assert r["mean"] < mean_tol                  # ... and the oracle reproduces it
    assert r["p99"] <= p99_tol
    # the rank decisions and spectra are the committed ones (guards against LAPACK / numpy drift of the oracle)
    assert cuts == [(c["n"], c["kept"]) for c in g["cuts"]]
    assert r["S"].size == g["K_out"]
    np.testing.assert_allclose(r["S"], g["eigvals"], rtol=1e-6)
    np.testing.assert_allclose(r["layer_norms"], g["layer_norms"], rtol=1e-5)


def test_most_readme_pairs_are_rank_truncated(golden):
    """the regime statement of the module docstring, from the committed numbers"""
    trunc = [n for n, g in golden.items() if any(c["kept"] < c["n"] for c in g["cuts"])]
    at_cut = [n for n, g in golden.items()
              if any(c["first_dropped"] is not None and c["last_kept"] < 2e-10 for c in g["cuts"])]
    assert len(trunc) >= 8 and "flower" not in trunc
    assert len(at_cut) >= 7
