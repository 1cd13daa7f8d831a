#!/usr/bin/env python3
"""Measured HBM rates on this box (SURVEY.md section 8d asks for measured peaks beside the vendor figure):
device-to-device copy (read + write), a write-only fill and a read-only reduction over buffers far larger than
the 256 MB of Infinity Cache."""
import json
import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = 1 << 30  # 4 GiB of fp32 per buffer
    a = torch.rand(n, device="cuda", dtype=torch.float32)
    b = torch.empty_like(a)
    nbytes = a.numel() * 4
    res = {}
    ms = timed(lambda: b.copy_(a))
    res["copy_GBs_read_plus_write"] = 2 * nbytes / (ms * 1e-3) / 1e9
    ms = timed(lambda: b.fill_(1.0))
    res["fill_GBs_write_only"] = nbytes / (ms * 1e-3) / 1e9
    ms = timed(lambda: a.sum())
    res["sum_GBs_read_only"] = nbytes / (ms * 1e-3) / 1e9
    c = torch.empty_like(a)
    ms = timed(lambda: torch.add(a, b, out=c))
    res["add_GBs_two_reads_one_write"] = 3 * nbytes / (ms * 1e-3) / 1e9
    print(json.dumps({"buffer_GiB": nbytes / 2**30, **{k: round(v, 1) for k, v in res.items()}}))


if __name__ == "__main__":
    main()
