"""Pins the CPU oracle where the reference actually operates: every sample pair of the reference's README table
(README.md:72-83: input image, enhance arguments, the author's output image; fixtures under tests/golden/).  Most of
the eleven pairs are rank-truncated -- eigenvalues of Ka, Wa and Q sit right at the 1e-10 cut (src/filter.cpp:214 via
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
# rock2 (README.md:82, "requires a lot of memory ... consider downsampling"): the cut on Wa drops 200 of 592 eigenvalues, the
# same amplification of the unpinned 8-bit Lab rounding as on bird; measured mean 2.02, p99 17
TOL["rock2"] = (2.5, 20.0)
MOVED_MIN = {"paper": 4.5}   # how far the edit moves L at least (mean grey levels); 5 elsewhere

_cache = {}


def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_readme_golden", os.path.join(GOLDEN, "make_readme_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def oracle_run(oracle, name, stored_ok=True):
    """what the oracle computes on a pair.  For the pairs of make_readme_golden.STORED (minutes and gigabytes on the CPU)
    the committed record of that run, unless `stored_ok` is False; the record has no `layers` but `layer_probes`, every
    `probe_step`-th pixel of every layer."""
    mod = _golden_module()
    if stored_ok and name in mod.STORED:
        return mod.load_stored(name)
    if name not in _cache:
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
    if name in _golden_module().STORED and not os.environ.get("NLE_TEST_BIG_PAIRS"):
        r = oracle_run(oracle, name)   # the committed record of the run (NLE_TEST_BIG_PAIRS=1 repeats the run itself)
        r["layers"] = None
    else:
        r = oracle_run(oracle, name, stored_ok=False)
    g = golden[name]
    cuts = [(c["n"], c["kept"]) for c in r["info"]]
    print(f"{name}: p = {cuts[0][0]}, kept by the cuts Ka/Wa/Q = {[c[1] for c in cuts]}, K' = {r['S'].size}, "
          f"lambda at the cuts = {[(c['last_kept'], c['first_dropped']) for c in r['info']]}; filter moves L by "
          f"{r['moved']:.2f}; oracle vs author's output: mean |dL| = {r['mean']:.3f}, p99 = {r['p99']:.1f}")
    mean_tol, p99_tol = TOL[name]
    assert r["moved"] > MOVED_MIN.get(name, 5.0)  # the edit is large ...
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
