#!/usr/bin/env python3
"""PCIe-inclusive end-to-end time at cfg4: nle_train_host + nle_apply_layers_host (host fp32 plane in, L host
planes out) next to the HBM-resident figure bench.py reports as `value`."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
    H, W, L = cfg["H"], cfg["W"], cfg["L"]
    lib = nle.lib()
    ctx = nle.Context(0)
    lum = np.ascontiguousarray(synth.synthetic_luminance(H, W), dtype=np.float32)
    out = np.empty((L, H * W), dtype=np.float32)
    times = []
    for it in range(5):
        f = C.c_void_p()
        t0 = time.perf_counter()
        st = lib.nle_train_host(ctx._h, lum.ctypes.data_as(C.c_void_p), H, W, cfg["n_row"], cfg["n_col"],
                                float(cfg["hx"]), float(cfg["hy"]), cfg["T"], cfg["K"], C.byref(f))
        assert st == 0, lib.nle_last_error(ctx._h)
        st = lib.nle_apply_layers_host(f, lum.ctypes.data_as(C.c_void_p), H, W, L, out.ctypes.data_as(C.c_void_p))
        assert st == 0
        times.append(time.perf_counter() - t0)
        lib.nle_filter_destroy(f)
    t = float(np.median(times[1:]))
    print(json.dumps({"workload": f"{H}x{W}", "pcie_inclusive_ms": t * 1e3, "pcie_inclusive_MPs": H * W / 1e6 / t,
                      "bytes_up": lum.nbytes * 2, "bytes_down": out.nbytes, "note": "pageable host buffers"}))


if __name__ == "__main__":
    main()
