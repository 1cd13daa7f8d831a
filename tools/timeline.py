#!/usr/bin/env python3
"""Timeline of the last train + apply step in a rocprofv3 --kernel-trace CSV: every kernel with its start (ms from the step's
first kernel), duration and the idle gap in front of it; gaps above a threshold are what the GPU spent waiting for the host.

    python tools/timeline.py <dir-or-csv> [gap_us=50] [--all]
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("nlek::", "")


def main():
    path = sys.argv[1]
    gap_us = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 50.0
    show_all = "--all" in sys.argv
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_gather_samples") and not r[2].endswith("_slab")]
    first = starts[-1] if starts else 0
    step = rows[first:]
    t0 = step[0][0]
    busy = 0
    prev_end = t0
    agg = {}
    print(f"# {path}: {len(step)} kernels in the last step")
    for s, e, n in step:
        gap = (s - prev_end) / 1e3
        busy += e - s
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
        if show_all or gap >= gap_us:
            print(f"t={(s - t0) / 1e6:9.3f} ms  gap {gap:9.1f} us  then {n} ({(e - s) / 1e3:.1f} us)")
        prev_end = max(prev_end, e)
    total = (prev_end - t0) / 1e6
    print(f"# span {total:.3f} ms, kernels busy {busy / 1e6:.3f} ms, idle {total - busy / 1e6:.3f} ms")
    for n, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"#   {ns / 1e6:9.3f} ms  {cnt:4d} x {ns / cnt / 1e3:9.1f} us  {n}")


if __name__ == "__main__":
    main()
