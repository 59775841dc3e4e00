#!/usr/bin/env python3
"""Summarise a COX_TIMELINE dump (lines "class start_ms end_ms", HIP events of every timed region relative to the moment
profiling was switched on): duration per kernel class, frame period, depth of the pipeline, and a few frames as they ran.

    COX_TIMELINE=out.txt python bench.py --steps 60 --warmup 20 ...;  python scripts/stage_timeline.py out.txt
A region's START event sits behind the stream's cross-stream waits, so it can be early by the wait; END events are exact."""
import sys

NAMES = ("merge", "apply", "bundle_hash", "point_sort", "touch_emit", "record_sort", "fast_start", "fast_visits", "fast_sweeps")
ORDER = ("bundle_hash", "point_sort", "merge", "touch_emit", "record_sort", "apply")
rows = [l.split() for l in open(sys.argv[1]) if l.strip() and not l.startswith("#")]
by = {}
for c, a, b in rows:
    by.setdefault(NAMES[int(c)], []).append((float(a), float(b)))
keep = 40
print(f"# last {keep} timed frames; times in us")
print(f"{'class':12s} {'regions':>7s} {'mean duration':>14s} {'period (end to end)':>20s}")
for name in ORDER:
    v = sorted(by.get(name, []))[-keep:]
    if len(v) < 2:
        continue
    print(f"{name:12s} {len(v):7d} {1e3 * sum(b - a for a, b in v) / len(v):14.1f} {1e3 * (v[-1][1] - v[0][1]) / (len(v) - 1):20.1f}")
if "bundle_hash" in by and "apply" in by:
    h, ap = sorted(by["bundle_hash"]), sorted(by["apply"])
    n = min(len(h), len(ap))
    lat = [ap[i][1] - h[i][0] for i in range(max(0, n - keep), n)]
    per = (ap[n - 1][1] - ap[max(0, n - keep)][1]) / max(1, min(keep, n) - 1)
    print(f"frame latency (bundle_hash start -> apply end of the same frame): {1e3 * sum(lat) / len(lat):.1f} us = {sum(lat) / len(lat) / per:.1f} frame periods in flight")
    i0 = max(0, n - 6)
    t0 = h[i0][0]
    print("# the last frames as they ran (end times; frame k's stages in columns):")
    print("frame " + " ".join(f"{x:>12s}" for x in ORDER))
    for i in range(i0, n):
        cells = []
        for name in ORDER:
            v = sorted(by[name])
            cells.append(f"{1e3 * (v[i][1] - t0):12.1f}" if i < len(v) else " " * 12)
        print(f"{i - i0:5d} " + " ".join(cells))
