"""Build recipe: hipcc cross-compiles the gfx950 shared library in-tree (no GPU needed).

    python nonlocal-image-edit_amd/build.py        # builds lib/libnle_hip.so (+ bin/enhance)

Outputs (git-ignored, but they travel to the GPU box with the snapshot):
    nonlocal-image-edit_amd/lib/libnle_hip.so   HIP kernels + host pipeline + C ABI (include/nle.h)
    nonlocal-image-edit_amd/bin/enhance         C++ CLI with the reference's argv (host/enhance.cpp)
    nonlocal-image-edit_amd/bin/test_filter     C++ port of the reference's unit tests over the C++ surface
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIBDIR = os.path.join(HERE, "lib")
BINDIR = os.path.join(HERE, "bin")
LIB = os.path.join(LIBDIR, "libnle_hip.so")

LIB_SOURCES = ["kernels.hip", "tsgemm_bf16x3.hip", "fused.hip", "sorted.hip", "generic64.hip", "tridiag.hip", "dense64.hip", "colour.hip", "pipeline.hip", "ortho.hip", "abi_ctx.hip", "devsolve.hip", "eigen_sym.cpp", "lab8_tables.cpp"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _includes(path: str, seen=None) -> set:
    """the quoted headers a source pulls in, transitively (paths relative to the including file)"""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        h = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if h not in seen and os.path.exists(h):
            seen.add(h)
            _includes(h, seen)
    return seen


def build_lib(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + sorted(_includes(s))):
            if s.endswith(".cpp"):  # host-only fp64 algebra: plain g++ (needs function multiversioning)
                # lab8_tables.cpp reproduces single-precision table arithmetic operation by operation: no contraction
                extra = ["-ffp-contract=off"] if os.path.basename(s) == "lab8_tables.cpp" else []
                jobs.append(["g++", "-O3", "-fopenmp-simd", "-std=c++17", "-fPIC", "-pthread", "-Wno-psabi"] + extra +
                            ["-I", os.path.join(ROOT, "include"), "-c", s, "-o", o])
            else:
                jobs.append([_hipcc(), "-x", "hip", "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC",
                             "-I", os.path.join(ROOT, "include"), "-c", s, "-o", o])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(os.environ.get("NLE_BUILD_JOBS", "4"))) as ex:
            list(ex.map(_run, jobs))
    if jobs or force or _newer(LIB, objs):
        _run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lpthread", "-ldl"])
    return LIB


def build_host(force: bool = False):
    """C++ surface (filter.hpp names) + enhance CLI + unit-test binary, linked to the .so."""
    os.makedirs(BINDIR, exist_ok=True)
    out = []
    common = [os.path.join(HOST, "filter.cpp"), os.path.join(HOST, "image_io.cpp"), os.path.join(HOST, "jpeg.cpp")]
    hdrs = [os.path.join(ROOT, "include", "nle", "filter.hpp"), os.path.join(ROOT, "include", "nle", "image_io.hpp"),
            os.path.join(ROOT, "include", "nle.h"), os.path.join(HOST, "cli_common.hpp")]
    for name, main in (("enhance", "enhance.cpp"), ("denoise", "denoise.cpp"), ("test_filter", "test_filter.cpp")):
        target = os.path.join(BINDIR, name)
        srcs = common + [os.path.join(HOST, main)]
        if not all(os.path.exists(s) for s in srcs):
            continue
        if force or _newer(target, srcs + hdrs + [LIB]):
            _run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include")] + srcs +
                 ["-L", LIBDIR, "-lnle_hip", "-Wl,-rpath,$ORIGIN/../lib", "-o", target])
        out.append(target)
    return out


def gen_abi():
    """the ctypes mirror's signature table, regenerated from include/nle.h"""
    import subprocess
    import sys
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ctypes.py")], check=True)


def build_all(force: bool = False):
    gen_abi()
    lib = build_lib(force)
    bins = build_host(force)
    return lib, bins


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
