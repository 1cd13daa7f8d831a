#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory into profiles/<tag>_{kernel_stats,hbm_traffic}.csv.

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB and are
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts wide coalesced streaming reads at
exactly 1/2 of the bytes moved (128-B requests tallied at 64 B), so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import os
import shutil
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for kind, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    agg[r["Kernel_Name"]][ctr].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_avg_raw", "WRITE_SIZE_KiB_avg_raw",
                    "read_bytes_per_launch_corrected(x2)", "write_bytes_per_launch", "hbm_bytes_per_launch"])
        for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
            fs = d.get("FETCH_SIZE", [0.0])
            ws = d.get("WRITE_SIZE", [0.0])
            f_avg, w_avg = sum(fs) / len(fs), sum(ws) / len(ws)
            rd, wr = 2.0 * f_avg * 1024, w_avg * 1024
            w.writerow([k, len(fs), f"{f_avg:.1f}", f"{w_avg:.1f}", f"{rd:.0f}", f"{wr:.0f}", f"{rd + wr:.0f}"])
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
