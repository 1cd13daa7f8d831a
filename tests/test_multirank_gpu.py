"""Row-slab sharding on the real kernels: 2 and 3 ranks share cuda:0 (one process each, `gloo`
all-reduce of the CUDA fp64 buffers -- RCCL cannot put two ranks on one device) and must
reproduce the single-rank result: only the order of the fp64 partial sums differs."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_l2

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, args, mode, outdir, slabs=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W, nr, nc, hx, hy, T, K, L = args
        x = synth.synthetic_luminance(H, W).astype(np.float32)
        ctx = nle.Context(0)
        ctx.set_mode(mode)
        g = nle.sample_grid(H, W, nr, nc)
        ctx.set_shard(rank, world, g["n_sel_rows"] * g["n_sel_cols"], lambda t: dist.all_reduce(t))
        if slabs:      # every rank hands over (and uploads) only its own rows: nle_ctx_set_slab_input
            ctx.set_slab_input(True)
            r0, r1 = nle.slab_rows(H, rank, world)
            xs = np.ascontiguousarray(x[r0:r1])
            if slabs == "host":
                f = nle.NLEFilter(ctx).train_filter_host(xs, nr, nc, hx, hy, T, K, shape=(H, W))
                Y = np.empty((L, (r1 - r0) * W), dtype=np.float32)
                f.apply_layers_host(None, L, Y)
                Y2 = np.empty_like(Y)
                f.apply_layers_host(xs, L, Y2)
                assert np.array_equal(Y, Y2)
            else:
                f = nle.NLEFilter(ctx).train_filter(xs, nr, nc, hx, hy, T, K, shape=(H, W))
                Y = f.apply_layers(xs, L).cpu().numpy()
        else:
            f = nle.NLEFilter(ctx).train_filter(x, nr, nc, hx, hy, T, K)
            Y = f.apply_layers(x, L).cpu().numpy()
        info = f.info()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), Y=Y, S=f.eigvals, rows=np.array([info["row0"], info["row1"]]))
        f.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", [1, 2, 3], ids=["materialised", "phi_free", "phi_free_exp"])
def test_sharded_ranks_match_single_rank(nle, oracle, ctx, tmp_path, world, mode):
    import torch.multiprocessing as mp
    args = (96, 128, 6, 8, 32.0, 30.0, 10, 10, 4)
    H, W, nr, nc, hx, hy, T, K, L = args
    mp.spawn(_worker, args=(world, _free_port(), args, mode, str(tmp_path)), nprocs=world, join=True)
    x = oracle.synthetic_luminance(H, W)
    ctx.set_mode(mode)
    try:
        f = nle.NLEFilter(ctx).train_filter(x.astype(np.float32), nr, nc, hx, hy, T, K)
        Y1 = f.apply_layers(x.astype(np.float32), L).cpu().numpy().astype(np.float64)
        S1 = f.eigvals
    finally:
        ctx.set_mode(0)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert parts[0]["rows"][0] == 0 and parts[-1]["rows"][1] == H
    for a, b in zip(parts[:-1], parts[1:]):
        assert a["rows"][1] == b["rows"][0]
    Y = np.concatenate([p["Y"] for p in parts], axis=1).astype(np.float64)
    for p in parts:
        assert rel_l2(p["S"], S1) < 1e-9
    for j in range(L):
        assert rel_l2(Y[j], Y1[j]) < 1e-6, j     # G = 1 vs G > 1 within 1e-6
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    for j in range(L):
        assert rel_l2(Y[j], Y_o[j]) < 1e-4, j


@pytest.mark.parametrize("how", ["device", "host"])
@pytest.mark.parametrize("mode", [0, 4, 1], ids=["tables", "materialised_f64", "materialised"])
def test_slab_input_matches_full_plane_input(nle, tmp_path, mode, how):
    """nle_ctx_set_slab_input: ranks that hand over only their own rows (device planes, or host buffers of which only
    n_local * 4 bytes are uploaded) get bit for bit what they get when every rank holds the full plane: the sample
    values and x at the samples arrive through an all-reduce of exact values (one non-zero contribution each)."""
    import torch.multiprocessing as mp
    args = (96, 128, 6, 8, 32.0, 30.0, 6, 10, 4)
    H, W, nr, nc, hx, hy, T, K, L = args
    world = 3
    full, slabs = tmp_path / "full", tmp_path / "slabs"
    full.mkdir()
    slabs.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), args, mode, str(full)), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), args, mode, str(slabs), how), nprocs=world, join=True)
    for r in range(world):
        a, b = np.load(full / f"rank{r}.npz"), np.load(slabs / f"rank{r}.npz")
        assert np.array_equal(a["rows"], b["rows"])
        assert np.array_equal(a["S"], b["S"]), (r, np.abs(a["S"] - b["S"]).max())
        assert np.array_equal(a["Y"], b["Y"]), (r, np.abs(a["Y"] - b["Y"]).max())


@pytest.mark.gpu
def test_native_rccl_single_rank(nle, oracle):
    """the library's own RCCL path (nle_rccl_unique_id / nle_ctx_init_rccl: librccl.so loaded on demand, ncclAllReduce
    in place on the ctx's stream) at world = 1 -- every all-reduce site of train and apply goes through it -- must
    reproduce the run without a communicator bit for bit.  (More ranks need more GPUs: two ranks cannot share one
    device in an RCCL communicator; the row-slab logic itself is covered above and in tests/test_sharding_gloo.py.)"""
    H, W, nr, nc, hx, hy, T, K, L = 96, 128, 6, 8, 32.0, 30.0, 6, 10, 4
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    c0 = nle.Context(0)
    f0 = nle.NLEFilter(c0).train_filter(x, nr, nc, hx, hy, T, K)
    Y0 = f0.apply_layers(x, L).cpu().numpy()
    ev0 = f0.eigvals.copy()
    f0.close()
    c0.close()
    c1 = nle.Context(0)
    uid = nle.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    c1.init_rccl(0, 1, uid)
    for mode in (0, 4):
        c1.set_mode(mode)
        f1 = nle.NLEFilter(c1).train_filter(x, nr, nc, hx, hy, T, K)
        Y1 = f1.apply_layers(x, L).cpu().numpy()
        if mode == 0:
            assert np.array_equal(f1.eigvals, ev0) and np.array_equal(Y1, Y0)
        else:
            assert rel_l2(Y1, Y0) < 1e-6
        f1.close()
    c1.close()


@pytest.mark.gpu
def test_native_rccl_bad_arguments_leave_the_ctx_usable(nle, oracle):
    """failure injection at world = 1: every malformed bootstrap call is refused with NLE_ERR_INVALID
    before anything collective is started, and the ctx filters afterwards exactly as one that never saw them"""
    import ctypes as C
    H, W, nr, nc, hx, hy, T, K, L = 64, 96, 5, 6, 28.0, 30.0, 5, 8, 3
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    c0 = nle.Context(0)
    f0 = nle.NLEFilter(c0).train_filter(x, nr, nc, hx, hy, T, K)
    Y0 = f0.apply_layers(x, L).cpu().numpy()
    f0.close()
    c = nle.Context(0)
    uid = nle.rccl_unique_id()
    for rank, world in ((1, 1), (-1, 1), (0, 0), (2, 2)):
        with pytest.raises(nle.NLEError) as ei:
            c.init_rccl(rank, world, uid)
        assert ei.value.code == nle.NLE_ERR_INVALID, (rank, world)
    L_ = nle.lib()
    short = (C.c_char * 64)()
    assert L_.nle_ctx_init_rccl(c._h, 0, 1, short, 64) == nle.NLE_ERR_INVALID   # id buffer too small
    assert L_.nle_ctx_init_rccl(c._h, 0, 1, None, 128) == nle.NLE_ERR_INVALID
    assert L_.nle_ctx_init_rccl(None, 0, 1, short, 128) == nle.NLE_ERR_INVALID
    assert L_.nle_rccl_unique_id(short, 64) == nle.NLE_ERR_INVALID
    assert L_.nle_ctx_set_rccl_comm(c._h, 0, 1, None) == nle.NLE_ERR_INVALID
    f = nle.NLEFilter(c).train_filter(x, nr, nc, hx, hy, T, K)
    assert np.array_equal(f.apply_layers(x, L).cpu().numpy(), Y0)
    f.close()
    c.close()
    c0.close()


@pytest.mark.gpu
def test_native_rccl_reinit_and_lifetimes(nle, oracle):
    """communicator lifetime at world = 1: a second nle_ctx_init_rccl on the same ctx replaces (and destroys) the first
    communicator; a ctx with a communicator can be destroyed while another one lives; a filter trained before the
    re-initialisation still applies; the next ctx bootstraps again from a fresh id.  All outputs bitwise equal."""
    H, W, nr, nc, hx, hy, T, K, L = 64, 96, 5, 6, 28.0, 30.0, 5, 8, 3
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    ca = nle.Context(0)
    ca.init_rccl(0, 1, nle.rccl_unique_id())
    fa = nle.NLEFilter(ca).train_filter(x, nr, nc, hx, hy, T, K)
    Ya = fa.apply_layers(x, L).cpu().numpy()
    ca.init_rccl(0, 1, nle.rccl_unique_id())          # replaces the communicator under a live filter
    assert np.array_equal(fa.apply_layers(x, L).cpu().numpy(), Ya)
    cb = nle.Context(0)
    cb.init_rccl(0, 1, nle.rccl_unique_id())          # two communicators of one process on one device, one after the other
    fb = nle.NLEFilter(cb).train_filter(x, nr, nc, hx, hy, T, K)
    fa.close()
    ca.close()                                        # destroy order: the older ctx first, the younger keeps working
    assert np.array_equal(fb.apply_layers(x, L).cpu().numpy(), Ya)
    fb.close()
    cb.close()
    cc = nle.Context(0)
    cc.init_rccl(0, 1, nle.rccl_unique_id())
    fc = nle.NLEFilter(cc).train_filter(x, nr, nc, hx, hy, T, K)
    assert np.array_equal(fc.apply_layers(x, L).cpu().numpy(), Ya)
    fc.close()
    cc.close()


def _fault_worker(rank, world, port, mode, outdir):
    """rank 1 is made to answer "Phi does not fit here": what do BOTH ranks do?  The fault is injected from here, through
    the all-reduce callback -- the agreement of a rank-local verdict (ranks_where) is the library's only ONE-double
    all-reduce, and rank 1 adds 1 to its contribution -- not by a hook in the product."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as entry
    nle = entry.load_package()
    synth = entry._load("nle_amd_synthetic", os.path.join(entry.PKG_DIR, "synthetic.py"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W, nr, nc, hx, hy, T, K, L = 96, 128, 6, 8, 32.0, 30.0, 6, 10, 3
        x = synth.synthetic_luminance(H, W).astype(np.float32) + (0.25 if mode == 0 else 0.0)   # mode 0: a non-integer plane
        ctx = nle.Context(0)
        ctx.set_mode(mode)
        g = nle.sample_grid(H, W, nr, nc)
        def allreduce(t):
            if t.numel() == 1 and rank == 1:
                t += 1.0            # "it does not fit HERE"
            dist.all_reduce(t)

        ctx.set_shard(rank, world, g["n_sel_rows"] * g["n_sel_cols"], allreduce)
        try:
            f = nle.NLEFilter(ctx).train_filter(x, nr, nc, hx, hy, T, K)
            Y = f.apply_layers(x, L).cpu().numpy()
            np.savez(os.path.join(outdir, f"rank{rank}.npz"), Y=Y, form=f.diag()["formulation"], code=0)
            f.close()
        except nle.NLEError as e:
            np.savez(os.path.join(outdir, f"rank{rank}.npz"), code=e.code)
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_a_refusal_on_one_rank_is_a_refusal_on_all(nle, tmp_path):
    """a rank that finds Phi does not fit used to throw alone and leave its peer in the next all-reduce for
    ever.  The verdict is now agreed first (ranks_where): with rank 1 forced to say "does not fit", BOTH ranks return
    NLE_ERR_INVALID from the materialised fp64 mode -- and the run ends."""
    import torch.multiprocessing as mp
    mp.spawn(_fault_worker, args=(2, _free_port(), nle.MODE_MATERIALISED_F64, str(tmp_path)), nprocs=2, join=True)
    codes = [int(np.load(tmp_path / f"rank{r}.npz")["code"]) for r in range(2)]
    assert codes == [nle.NLE_ERR_INVALID, nle.NLE_ERR_INVALID], codes


def test_one_rank_short_of_memory_makes_every_rank_stream(nle, tmp_path):
    """auto mode's fp64 fallback: the choice between holding Phi and the streamed form depends on the free memory of each
    rank; the ranks agree on it (if one must stream, all do), so their collectives still pair up"""
    import torch.multiprocessing as mp
    mp.spawn(_fault_worker, args=(2, _free_port(), nle.MODE_AUTO, str(tmp_path)), nprocs=2, join=True)
    recs = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    assert [int(r["code"]) for r in recs] == [0, 0]
    assert [int(r["form"]) for r in recs] == [nle.MODE_STREAMED_F64] * 2
    assert all(np.isfinite(r["Y"]).all() for r in recs)


@pytest.mark.gpu
def test_aborted_communicator_fails_collectives_until_rebound(nle, oracle):
    """nle_ctx_abort_rccl (what a multi-rank host calls on every ctx when one rank has failed, so that the others' pending
    collectives end): afterwards a train on the ctx returns NLE_ERR_COMM instead of entering a collective, a second abort
    is a no-op, and nle_ctx_init_rccl binds a fresh communicator that works"""
    H, W, nr, nc, hx, hy, T, K, L = 64, 96, 5, 6, 28.0, 30.0, 5, 8, 3
    x = oracle.synthetic_luminance(H, W).astype(np.float32)
    c = nle.Context(0)
    c.abort_rccl()                                   # no communicator: nothing to do
    f = nle.NLEFilter(c).train_filter(x, nr, nc, hx, hy, T, K)
    Y0 = f.apply_layers(x, L).cpu().numpy()
    f.close()
    c.init_rccl(0, 1, nle.rccl_unique_id())
    c.abort_rccl()
    c.abort_rccl()
    with pytest.raises(nle.NLEError) as ei:
        nle.NLEFilter(c).train_filter(x, nr, nc, hx, hy, T, K)
    assert ei.value.code == nle.NLE_ERR_COMM
    c.init_rccl(0, 1, nle.rccl_unique_id())
    f = nle.NLEFilter(c).train_filter(x, nr, nc, hx, hy, T, K)
    assert np.array_equal(f.apply_layers(x, L).cpu().numpy(), Y0)
    f.close()
    c.close()


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_emits_the_multi_gpu_line(tmp_path):
    """The driver runs `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` on an 8-GPU node that this
    round never sees.  Rehearsal on ONE GPU: two ranks over gloo, both on cuda:0 (--same-device; the all-reduces go through
    the torch.distributed callback instead of RCCL, which cannot put two ranks on one device), on cfg2 so that it takes
    seconds.  Asserts the plumbing the first real run depends on: exit code 0, exactly one JSON line from rank 0, the bench
    contract's keys, the N > 1 legs (`shard_check` against the single-GPU result, `replicas`, slab-input `host_to_host`
    with its one-plane figure) and `value` = pixels of the whole image over the slowest rank's time."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--config", "cfg2", "--backend", "gloo", "--same-device", "--h2h-runs", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "host_to_host", "replicas", "shard_check", "rccl_ranks",
              "comm", "slab_input", "kernels"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "MP/s" and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "cfg2" in d["config"]["workload"] and "row-slab x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 512 * 512 / 1e6 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["shard_check"]["matches"] is True and d["shard_check"]["max_rel_l2_per_layer_vs_single_gpu"] <= 1e-6
    assert d["slab_input"] is True and d["rccl_ranks"] == 0 and "callback" in d["comm"]          # gloo rehearsal: no RCCL
    assert d["cpu_baseline"] is None                                                             # N = 1 only
    assert d["replicas"]["scaling"] == "weak" and d["replicas"]["value"] > 0
    h = d["host_to_host"]
    assert h["matches_device_resident_output"] is True and h["value"] > 0 and h["u8_plane"]["value"] > 0
    assert h["u8_plane"]["bytes_d2h"] * 4 * 4 == h["bytes_d2h"]                                  # one byte per pixel instead of L floats
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1.5
