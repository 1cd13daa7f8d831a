#!/bin/bash
# usage: tools/gap_probe.sh <cfg>  -- what the GPU does between two steps (kernel + memory-copy trace around step starts)
set -o pipefail
CFG=${1:-cfg5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/gap_$CFG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/trace_run.py $CFG > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(os.path.join(out, "trace", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
starts = [i for i, e in enumerate(ev) if "k_gather_samples" in e[2] and "slab" not in e[2]]
for si in starts[-2:]:
    t0 = ev[si][0]
    print("---- step start")
    for s, e, n in ev[max(0, si - 6):si + 8]:
        print(f"  t={(s - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e3:9.1f} us  {n}")
PY
find $OUT/trace -name '*.csv' -delete
