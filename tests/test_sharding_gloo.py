"""N > 1 path on CPU: two `gloo` ranks run the row-slab decomposition with REAL
torch.distributed all-reduces (the same three exchange steps the GPU path has: Sinkhorn column
sums, Gram partials, V^T x) through the streaming oracle, and must reproduce the single-process
result.  Also checks that the slab partition the C library hands each rank (nle_slab_rows) is the
one the decomposition assumes.  The GPU kernels themselves cannot run here; what this covers is
the sharding logic and the collective pattern (sum, fp64, r / r*r / K elements)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, rel_l2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, args, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    oracle = entry.load_oracle()
    nle = entry.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W, nr, nc, hx, hy, T, K, L = args
        x = oracle.synthetic_luminance(H, W)
        calls = []

        def allreduce(v):
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            calls.append(t.numel())
            return t.numpy().reshape(np.shape(v))

        r0, r1 = nle.slab_rows(H, rank, world)           # the C library's partition
        assert (r0, r1) == oracle.slab_rows(H, rank, world)
        V, S = oracle.train_filter_streaming(x, nr, nc, hx, hy, T, K, tile=1000, shard=(rank, world),
                                             allreduce=allreduce)
        Y = oracle.apply_layers_streaming(V, S, x.ravel()[r0 * W:r1 * W], L, allreduce=allreduce)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), Y=Y, S=S, rows=np.array([r0, r1]),
                 calls=np.array(calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("args", [(48, 64, 4, 5, 16.0, 30.0, 10, 8, 4), (33, 47, 3, 4, 20.0, 25.0, 3, 4, 2)])
def test_two_rank_gloo_matches_single_process(oracle, tmp_path, args):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), args, str(tmp_path)), nprocs=world, join=True)
    H, W, nr, nc, hx, hy, T, K, L = args
    x = oracle.synthetic_luminance(H, W)
    V_o, S_o = oracle.train_filter(x, nr, nc, hx, hy, T, K)
    Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert parts[0]["rows"][0] == 0 and parts[0]["rows"][1] == parts[1]["rows"][0] and parts[1]["rows"][1] == H
    Y = np.concatenate([p["Y"] for p in parts], axis=1)
    for j in range(L):
        assert rel_l2(Y[j], Y_o[j]) < 1e-6
    for p in parts:
        assert rel_l2(p["S"], S_o) < 1e-8
        calls = p["calls"].tolist()
        r = nr * nc
        # 2T+1 column-sum all-reduces of r doubles, one r x r Gram, one K-vector for apply
        assert calls.count(r) == 2 * T + 1 and calls.count(r * r) == 1 and calls[-1] == S_o.size
