"""The C++ drop-in surface (include/nle/filter.hpp, nonlocal-image-edit_amd/host/*.cpp) and the
`enhance` CLI (reference src/enhance.cpp): argv, stdout, exit codes, and -- on the GPU -- the
reference's unit tests through the C++ surface plus BASELINE.json configs[0] end to end through the
CLI against the README output image."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "nonlocal-image-edit_amd", "bin")
ENHANCE = os.path.join(BIN, "enhance")
TEST_FILTER = os.path.join(BIN, "test_filter")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not (os.path.exists(ENHANCE) and os.path.exists(TEST_FILTER)):
        import __graft_entry__ as entry
        entry.build()
    assert os.path.exists(ENHANCE) and os.path.exists(TEST_FILTER)


def test_cli_usage_goes_to_stderr_and_exits_zero():
    """src/enhance.cpp:15-18: fewer than 9 arguments -> usage on stderr, return 0"""
    r = subprocess.run([ENHANCE, "a.bmp", "b.bmp", "10", "20"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    assert r.stderr.startswith("Usage: ") and "<# sinkhorn iterations>" in r.stderr
    assert r.stdout == ""


def test_cli_unreadable_image_exits_zero(tmp_path):
    """src/enhance.cpp:34-37: imread failure -> message on stderr, return 0, nothing written"""
    out = tmp_path / "o.bmp"
    r = subprocess.run([ENHANCE, str(tmp_path / "missing.bmp"), str(out), "10", "20", "100", "30", "50", "30", "2", "3",
                        "4", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    assert "Failed to read file from" in r.stderr
    assert not out.exists()


def test_cli_garbage_number_throws_like_stoi():
    """std::stoi on garbage throws std::invalid_argument -> abnormal termination, like the reference"""
    r = subprocess.run([ENHANCE, "a.bmp", "b.bmp", "ten", "20", "100", "30", "50", "30", "2", "3", "4", "1"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0


@pytest.mark.gpu
def test_reference_unit_tests_through_cpp_surface():
    r = subprocess.run([TEST_FILTER], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 failed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["png", "bmp"])
def test_enhance_cli_flower_matches_readme_pair(oracle, tmp_path, ext):
    """BASELINE.json configs[0]: enhance flower-50.bmp <out> 10 20 100 30 50 30 2 3 4 1 (README.md:74)"""
    from PIL import Image
    out = tmp_path / f"flower-out.{ext}"
    r = subprocess.run([ENHANCE, os.path.join(GOLDEN, "flower-50.bmp"), str(out), "10", "20", "100", "30", "50", "30",
                        "2", "3", "4", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[:4] == ["Computing kernel", "Nystrom approximation", "Sinkhorn", "Orthogonalize"]  # :483-498
    assert sum(l.startswith("Eigvec ") and "minCoeff" in l for l in lines) == 5                      # :506
    assert lines[-1] == "Done. Press any key in result window to exit."
    got = np.asarray(Image.open(out).convert("RGB"))[..., ::-1]
    want = np.asarray(Image.open(os.path.join(GOLDEN, "flower-filtered.png")).convert("RGB"))[..., ::-1]
    assert got.shape == want.shape == (267, 400, 3)
    L_got = oracle.bgr_to_lab8(got)[..., 0].astype(np.float64)
    L_want = oracle.bgr_to_lab8(want)[..., 0].astype(np.float64)
    err = np.abs(L_got - L_want)
    print(f"CLI vs README output: mean |dL| = {err.mean():.3f}, p99 = {np.percentile(err, 99):.1f}")
    assert err.mean() < 0.2 and np.percentile(err, 99) <= 1.0
    # and against the oracle's own 8-bit L plane for config 1 (same Lab restatement on both sides)
    gold = np.load(os.path.join(GOLDEN, "flower_cfg1.npz"))
    src = np.asarray(Image.open(os.path.join(GOLDEN, "flower-50.bmp")).convert("RGB"))[..., ::-1]
    assert np.array_equal(oracle.bgr_to_lab8(src)[..., 0], gold["L_in"])
    d = np.abs(L_got - gold["L_out"].astype(np.float64))
    assert d.mean() < 0.6   # Lab -> BGR -> Lab round trip of the 8-bit image costs a fraction of a level


@pytest.mark.gpu
def test_enhance_cli_on_a_large_ppm_matches_the_python_pipeline(nle, oracle, ctx, tmp_path):
    """1536x1024 colour image through the CLI (device colour conversion + hot path) against the same
    steps done with the numpy Lab restatement and the ctypes mirror."""
    H, W = 1024, 1536
    base = oracle.synthetic_luminance(H, W)
    rr, cc = np.mgrid[0:H, 0:W]
    img = np.stack([np.clip(base * 0.8 + 20 * np.sin(cc / 97.0), 0, 255),
                    np.clip(base, 0, 255),
                    np.clip(base * 0.9 + 25 * np.cos(rr / 61.0), 0, 255)], axis=-1).astype(np.uint8)   # BGR
    src = tmp_path / "in.ppm"
    with open(src, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (W, H))
        fh.write(np.ascontiguousarray(img[..., ::-1]).tobytes())
    out = tmp_path / "out.ppm"
    args = ["8", "12", "300", "30", "10", "20", "2", "3", "4", "1"]
    r = subprocess.run([ENHANCE, str(src), str(out)] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    assert raw.startswith(b"P6\n1536 1024\n255\n")
    got = np.frombuffer(raw[len(b"P6\n1536 1024\n255\n"):], dtype=np.uint8).reshape(H, W, 3)[..., ::-1]
    lab = oracle.bgr_to_lab8(img)
    Lp = lab[..., 0].astype(np.float32)
    f = nle.NLEFilter(ctx).train_filter(Lp, 8, 12, 300.0, 30.0, 10, 20)
    y = f.apply(Lp, nle.transform_eigenvalues(f.eigvals, [2.0, 3.0, 4.0, 1.0])).cpu().numpy().reshape(H, W)
    lab2 = lab.copy()
    lab2[..., 0] = np.rint(np.clip(y.astype(np.float64), 0, 255)).astype(np.uint8)
    want = oracle.lab8_to_bgr(lab2)
    d = np.abs(got.astype(int) - want.astype(int))
    print("CLI vs python pipeline: max", d.max(), "mismatching values", (d > 0).mean())
    assert d.max() <= 3 and (d > 0).mean() < 1e-3     # rounding ties of the filtered L plane only: both conversions are exact integers
    f.close()


@pytest.mark.gpu
def test_enhance_cli_on_cfg5_s_3_channel_image(nle, oracle, ctx, tmp_path):
    """BASELINE.json configs[4] end to end: an 8192 x 8192 3-channel image through `enhance` with 30 x 30 samples, hx = W / 8,
    K = 100 and six weights (BGR -> Lab, train on L, apply, clamp / round, merge, Lab -> BGR -- all on the device) against the
    same steps done plane by plane: the oracle's integer colour conversions (numpy, on the CPU) around the filter run
    through the ctypes mirror.  Exact: both conversions are integer algorithms and the filter is the same library."""
    H = W = 8192
    base = oracle.synthetic_luminance(H, W)
    img = np.empty((H, W, 3), dtype=np.uint8)                                        # BGR
    for r0 in range(0, H, 1024):
        rr, cc = np.mgrid[r0:r0 + 1024, 0:W]
        b = base[r0:r0 + 1024]
        img[r0:r0 + 1024, :, 0] = np.clip(b * 0.8 + 20 * np.sin(cc / 397.0), 0, 255)
        img[r0:r0 + 1024, :, 1] = np.clip(b, 0, 255)
        img[r0:r0 + 1024, :, 2] = np.clip(b * 0.9 + 25 * np.cos(rr / 261.0), 0, 255)
    src, out = tmp_path / "in.ppm", tmp_path / "out.ppm"
    head = b"P6\n%d %d\n255\n" % (W, H)
    with open(src, "wb") as fh:
        fh.write(head)
        for r0 in range(0, H, 1024):
            fh.write(np.ascontiguousarray(img[r0:r0 + 1024, :, ::-1]).tobytes())
    args = ["30", "30", "1024", "30", "10", "100", "2", "3", "3", "4", "4", "1"]
    r = subprocess.run([ENHANCE, str(src), str(out)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    assert bytes(raw[:len(head)]) == head
    got = raw[len(head):].reshape(H, W, 3)                                           # RGB
    lab = np.concatenate([oracle.bgr_to_lab8(img[r0:r0 + 512]) for r0 in range(0, H, 512)])
    Lp = lab[..., 0].astype(np.float32)
    f = nle.NLEFilter(ctx).train_filter(Lp, 30, 30, 1024.0, 30.0, 10, 100)
    d = f.diag()
    assert d["formulation"] == nle.MODE_PHI_FREE and d["p"] == 900 and d["K"] == 100
    fs = nle.transform_eigenvalues(f.eigvals, [2.0, 3.0, 3.0, 4.0, 4.0, 1.0])
    lab[..., 0] = f.apply_u8(Lp, fs).cpu().numpy().reshape(H, W)
    f.close()
    ctx.trim()
    bad = worst = 0
    for r0 in range(0, H, 512):
        want = oracle.lab8_to_bgr(lab[r0:r0 + 512])[..., ::-1]
        dd = np.abs(got[r0:r0 + 512].astype(np.int16) - want.astype(np.int16))
        bad += int((dd > 0).sum())
        worst = max(worst, int(dd.max()))
    moved = float(np.abs(got[..., 1].astype(np.int16) - img[..., 1].astype(np.int16)).mean())
    print(f"cfg5 3-channel CLI vs plane-level pipeline: {bad} of {got.size} values differ, max {worst}; the edit moves G by {moved:.2f}")
    assert bad == 0 and moved > 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_enhance_cli_over_a_device_group_matches_one_device(tmp_path, devices):
    """NLE_DEVICES: `bin/enhance` shards the image by row slabs over one context and host thread per listed device
    (slab input).  With the same device listed several times the all-reduces go through a
    host-mediated sum in rank order (distinct devices use the library's RCCL communicator: needs more than one GPU);
    the written image must be the single-device one up to isolated 8-bit rounding ties, the banners identical."""
    from PIL import Image
    src = os.path.join(GOLDEN, "flower-50.bmp")
    args = ["10", "20", "100", "30", "50", "30", "2", "3", "4", "1"]
    outs, texts = [], []
    for tag, env in (("one", {}), ("group", {"NLE_DEVICES": devices})):
        out = tmp_path / f"{tag}.png"
        e = dict(os.environ)
        e.pop("NLE_DEVICES", None)
        e.update(env)
        r = subprocess.run([ENHANCE, src, str(out)] + args, capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, r.stderr
        outs.append(np.asarray(Image.open(out).convert("RGB")).astype(int))
        texts.append(r.stdout.splitlines())
    assert texts[0][:4] == texts[1][:4] and texts[0][-1] == texts[1][-1]
    ev = [[float(l.split()[3]) for l in t if l.startswith("Eigvec ")] for t in texts]
    assert len(ev[0]) == len(ev[1]) == 5 and np.allclose(ev[0], ev[1], rtol=1e-5)   # the banner prints 6 digits
    d = np.abs(outs[0] - outs[1])
    print("device group", devices, "vs one device: max", d.max(), "differing values", (d > 0).mean())
    assert d.max() <= 2 and (d > 0).mean() < 1e-3
