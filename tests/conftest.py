"""Shared test plumbing.

`nle_amd`  -- the Python mirror of the C ABI (package dir `nonlocal-image-edit_amd/`, whose
              name is not an identifier, so it is loaded by path under this alias).
`oracle`   -- the CPU oracle (test infrastructure; only tests/bench/smoke may import it).
"""
import importlib.util
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "nonlocal-image-edit_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_nle_amd():
    if "nle_amd" in sys.modules:
        return sys.modules["nle_amd"]
    spec = importlib.util.spec_from_file_location(
        "nle_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nle_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    if "nle_oracle" in sys.modules:
        return sys.modules["nle_oracle"]
    spec = importlib.util.spec_from_file_location("nle_oracle", os.path.join(ROOT, "oracle", "nle_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nle_oracle"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nle():
    return load_nle_amd()


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


@pytest.fixture(scope="session")
def ctx(nle):
    """One nle_ctx on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    c = nle.Context(0)
    yield c
    c.close()


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
