"""the denoise wrapper either side of the hot path (reference src/denoise.cpp,
src/filter.cpp:349-410, 521-538) -- the bilateral prefilter as a HIP kernel, NLEFilter::trainForDenoise / denoise
on the C++ surface and the `denoise` CLI.  OpenCV is not in this image, so `cv::bilateralFilter` is restated from its
documentation (oracle.bilateral8); agreement with an actual OpenCV build is unpinned, like the Lab conversion."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "nonlocal-image-edit_amd", "bin")
DENOISE = os.path.join(BIN, "denoise")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(DENOISE):
        import __graft_entry__ as entry
        entry.build()
    assert os.path.exists(DENOISE)


# ---------------------------------------------------------------- CPU: the oracle's restatement and the CLI shell
def test_oracle_bilateral_properties(oracle):
    rng = np.random.default_rng(3)
    flat = np.full((9, 13), 200)
    assert np.array_equal(oracle.bilateral8(flat, 10, 10), flat)            # constants are fixed points
    x = np.clip(128 + 60 * np.sin(np.arange(40)[None, :] / 5.0) + rng.integers(-10, 11, (30, 40)), 0, 255).astype(int)
    y = oracle.bilateral8(x, 25, 3)
    assert y.dtype == np.uint8 and y.shape == x.shape
    assert x.min() <= y.min() and y.max() <= x.max()                        # a convex combination of neighbours
    assert np.abs(np.diff(y.astype(int), axis=0)).mean() < np.abs(np.diff(x, axis=0)).mean()   # it smooths
    # an isolated spike far outside sigma_color keeps its value (its neighbours get ~zero colour weight) ...
    s = np.full((21, 21), 50)
    s[10, 10] = 250
    assert oracle.bilateral8(s, 5, 3)[10, 10] == 250
    # ... and does not leak into its neighbours
    assert oracle.bilateral8(s, 5, 3)[10, 11] == 50
    # radius and tables: d = -1 -> radius = round(1.5 sigma_space), circular support
    r, sw, cw = oracle.bilateral_tables(10, 10)
    assert r == 15 and sw.shape == (31, 31) and sw[0, 0] == 0 and sw[15, 0] > 0 and sw[15, 15] == 1 and cw[0] == 1


def test_oracle_reflect101(oracle):
    assert list(oracle._reflect101(np.arange(-3, 8), 5)) == [3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1]


def test_cli_usage_goes_to_stderr_and_exits_zero():
    """src/denoise.cpp:14-17: fewer than 11 arguments -> usage on stderr, return 0"""
    r = subprocess.run([DENOISE, "a.bmp", "b.bmp", "10", "20", "100", "30", "10", "20", "10", "10"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stderr.startswith("Usage: ") and r.stdout == ""


def test_cli_unreadable_image_exits_zero(tmp_path):
    """src/denoise.cpp:33-36"""
    out = tmp_path / "o.bmp"
    r = subprocess.run([DENOISE, str(tmp_path / "missing.bmp"), str(out), "10", "20", "100", "30", "10", "20", "10", "10", "2"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Failed to read file from" in r.stderr and not out.exists()


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("shape,sc,ss", [((96, 128), 10, 10), ((33, 47), 25, 3), ((7, 5), 10, 4), ((200, 64), 3, 1)])
def test_bilateral_kernel_is_bit_exact_against_the_oracle(nle, oracle, ctx, shape, sc, ss):
    rng = np.random.default_rng(11)
    H, W = shape
    rr, cc = np.mgrid[0:H, 0:W]
    x = np.clip(128 + 70 * np.sin(rr / 9.0 + cc / 13.0) + rng.integers(-15, 16, shape), 0, 255).astype(np.int64)
    want = oracle.bilateral8(x, sc, ss)
    got = ctx.bilateral8(x.astype(np.float32), sc, ss).cpu().numpy()
    assert np.array_equal(got, np.rint(got)) and got.min() >= 0 and got.max() <= 255
    assert np.array_equal(got.astype(np.uint8), want)


@pytest.mark.gpu
def test_bilateral_rejects_a_window_that_does_not_fit(nle, ctx):
    with pytest.raises(nle.NLEError, match="sigma_space"):
        ctx.bilateral8(np.zeros((16, 16), np.float32), 10, 50)


@pytest.mark.gpu
def test_channel_split_and_plane_merge(nle, oracle, ctx):
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    lab, L = ctx.bgr2lab8(bgr)
    for ch in range(3):
        assert np.array_equal(ctx.lab8_channel(lab, ch).cpu().numpy(), lab.cpu().numpy()[..., ch].astype(np.float32))
    # replacing a and b by their own float planes changes nothing; by out-of-range values it clamps
    a, b = ctx.lab8_channel(lab, 1), ctx.lab8_channel(lab, 2)
    assert np.array_equal(ctx.lab2bgr8(lab, L, a, b).cpu().numpy(), ctx.lab2bgr8(lab).cpu().numpy())
    lab_np = lab.cpu().numpy().copy()
    lab_np[..., 1] = 255
    lab_np[..., 2] = 0
    import torch
    want = ctx.lab2bgr8(torch.as_tensor(lab_np, device=lab.device)).cpu().numpy()
    assert np.array_equal(ctx.lab2bgr8(lab, None, a * 0 + 300.5, b * 0 - 7.25).cpu().numpy(), want)


@pytest.mark.gpu
def test_denoise_cli_matches_the_oracle_pipeline(oracle, tmp_path):
    """flower-50.bmp through `denoise` with the enhance README geometry (10 x 20 samples, hx 100, hy 30) and
    T = 10, K = 20, sigmaColor = sigmaSpace = 10, shrink 2, against oracle.denoise_image on the same bytes."""
    from PIL import Image
    src_path = os.path.join(GOLDEN, "flower-50.bmp")
    out = tmp_path / "flower-dn.png"
    args = ["10", "20", "100", "30", "10", "20", "10", "10", "2"]
    r = subprocess.run([DENOISE, src_path, str(out)] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[:4] == ["Computing kernel", "Nystrom approximation", "Sinkhorn", "Orthogonalize"]   # :483-498
    assert sum(l.startswith("eig ") and " val: " in l for l in lines) == 20                            # :383
    assert lines[-1] == "Done. Press any key in result window to exit."
    got = np.asarray(Image.open(out).convert("RGB"))[..., ::-1]
    src = np.asarray(Image.open(src_path).convert("RGB"))[..., ::-1]
    want = oracle.denoise_image(src, 10, 20, 100.0, 30.0, 10, 20, 10, 10, 2.0)
    assert got.shape == want.shape == src.shape
    d = np.abs(got.astype(int) - want.astype(int))
    print("denoise CLI vs oracle: max", d.max(), "mismatching values", (d > 0).mean())
    assert d.max() <= 3 and (d > 0).mean() < 5e-3          # rounding ties of the filtered a, b planes only
    # and it is a denoiser of the chroma: a/b channels get smoother, L is the bilateral-filtered plane
    lab_in, lab_out = oracle.bgr_to_lab8(src).astype(int), oracle.bgr_to_lab8(got).astype(int)
    for ch in (1, 2):
        assert np.abs(np.diff(lab_out[..., ch], axis=1)).mean() <= np.abs(np.diff(lab_in[..., ch], axis=1)).mean()


@pytest.mark.gpu
def test_bilateral8_refuses_a_plane_that_is_not_8_bit(nle, ctx):
    """the kernel indexes a 256-entry table with |v - v0|: a non-integer or out-of-range plane is an error, not an
    out-of-bounds LDS read; and NLEFilter.apply checks that f_s holds K' values before the ABI reads them"""
    x = np.full((32, 40), 17.5, dtype=np.float32)
    with pytest.raises(nle.NLEError):
        ctx.bilateral8(x, 10.0, 3.0)
    x[:] = 300.0
    with pytest.raises(nle.NLEError):
        ctx.bilateral8(x, 10.0, 3.0)
    y = np.random.default_rng(0).integers(0, 256, (48, 64)).astype(np.float32)
    f = nle.NLEFilter(ctx).train_filter(y, 4, 5, 16.0, 30.0, 3, 6)
    with pytest.raises(nle.NLEError):
        f.apply(y, np.ones(f.info()["K"] + 1))
    f.close()


@pytest.mark.gpu
def test_level_check_on_planes_that_are_not_16_byte_aligned(nle, ctx):
    """k_check_levels reads the plane with 16-byte loads between a scalar head and tail: a plane that starts 1, 2 or 3 floats
    off alignment (a row slab of an odd-width image does) is checked to its first and last pixel all the same"""
    import torch
    rng = np.random.default_rng(4)
    H, W = 37, 53                       # H * W is odd as well
    y = rng.integers(0, 256, (H, W)).astype(np.float32)
    want = ctx.bilateral8(y, 12.0, 2.0).cpu().numpy()
    for off in (1, 2, 3):
        buf = torch.zeros(H * W + 8, dtype=torch.float32, device="cuda:0")
        view = buf[off:off + H * W].view(H, W)
        view.copy_(torch.from_numpy(y))
        assert view.data_ptr() % 16 == 4 * off
        assert np.array_equal(ctx.bilateral8(view, 12.0, 2.0).cpu().numpy(), want)
        for pos in ((0, 0), (0, 2), (H - 1, W - 1), (H - 1, W - 3), (H // 2, W // 2)):
            keep = float(view[pos])
            view[pos] = keep + 0.5
            with pytest.raises(nle.NLEError):
                ctx.bilateral8(view, 12.0, 2.0)
            view[pos] = keep


@pytest.mark.gpu
def test_device_bgr2lab8_equals_the_oracle_on_every_colour(nle, oracle, ctx):
    """nle_bgr2lab8 is OpenCV's fixed-point 8-bit BGR -> Lab (integer arithmetic on the tables of nle_lab8_tables): all
    2^24 colours come out exactly as the oracle's restatement of that algorithm has them"""
    v = np.arange(256, dtype=np.uint8)
    for b in range(0, 256, 16):   # 16 blue values per slab: 16 x 256 x 256 colours
        cube = np.stack(np.meshgrid(v[b:b + 16], v, v, indexing="ij"), axis=-1).reshape(16 * 256, 256, 3)
        lab, L = ctx.bgr2lab8(cube)
        want = oracle.bgr_to_lab8(cube)
        assert np.array_equal(lab.cpu().numpy(), want), b
        assert np.array_equal(L.cpu().numpy(), want[..., 0].astype(np.float32))


@pytest.mark.gpu
def test_device_lab2bgr8_equals_the_oracle_on_every_lab_triple(nle, oracle, ctx):
    """nle_lab2bgr8 is OpenCV's integer 8-bit Lab -> BGR (Lab2RGBinteger on the tables of nle_lab8_inverse_tables; the
    inverse of f(t) computed in the kernel, looked up in the oracle): all 2^24 (L, a, b) triples come out exactly as the
    oracle's restatement of that algorithm has them -- and so do float planes handed in for L, a, b after their clamp and
    round-half-even (src/filter.cpp:434-436, :391-399)"""
    import torch
    v = np.arange(256, dtype=np.uint8)
    for l0 in range(0, 256, 16):   # 16 L values per slab: 16 x 256 x 256 triples
        cube = np.ascontiguousarray(np.stack(np.meshgrid(v[l0:l0 + 16], v, v, indexing="ij"), axis=-1).reshape(16 * 256, 256, 3))
        got = ctx.lab2bgr8(torch.as_tensor(cube, device="cuda:0")).cpu().numpy()
        assert np.array_equal(got, oracle.lab8_to_bgr(cube)), l0
    rng = np.random.default_rng(11)
    lab = rng.integers(0, 256, (64, 96, 3)).astype(np.uint8)
    Lf = rng.uniform(-20, 280, (64, 96)).astype(np.float32)
    Lf[0, :8] = [0.5, 1.5, 2.5, 254.5, 255.5, -0.5, 127.5, 128.5]          # ties go to even
    want = lab.copy()
    want[..., 0] = np.rint(np.clip(Lf.astype(np.float64), 0, 255)).astype(np.uint8)
    got = ctx.lab2bgr8(torch.as_tensor(lab, device="cuda:0"), torch.as_tensor(Lf, device="cuda:0")).cpu().numpy()
    assert np.array_equal(got, oracle.lab8_to_bgr(want))
