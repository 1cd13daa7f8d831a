"""Housekeeping: the product kernels carry no measurement-only code paths (the ablation / timestamp variants of
k_sorted_pass behind profiles/r2_pass_ablation.txt lived beside the sources as a patch until round 4; the kernel they
patched has changed since -- its pixel loop runs on moments now -- and the patch went with it: the profile is the record)."""
import glob
import os

from conftest import PKG_DIR


def test_product_sources_have_no_ablation_hooks():
    for f in glob.glob(os.path.join(PKG_DIR, "csrc", "*")) + glob.glob(os.path.join(PKG_DIR, "host", "*")):
        text = open(f, errors="replace").read()
        assert "NLE_ABL_" not in text and "NLE_STAMP" not in text, f
