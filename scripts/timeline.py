#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 kernel trace: busy time, average overlap, per-queue busy share, per-kernel mean duration
over the last `frac` of the run (the timed region of bench.py)."""
import csv, sys, glob, collections
path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = path if path.endswith(".csv") else glob.glob(path + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r["Queue_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
# window: the middle half of the fusion frames (between the 25 % and 75 % launch of the layer-update kernel), so that
# warm-up, the registration section and the CPU baseline of bench.py stay out of the statistics
marks = [r[0] for r in rows if r[2].startswith("k_apply_eval")]
if len(marks) >= 8:
    lo, hi = marks[len(marks) // 4], marks[3 * len(marks) // 4]
    rows = [r for r in rows if lo <= r[0] < hi]
else:
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; area = 0; cur = 0; last = ev[0][0]
hist = collections.Counter()
for t, d in ev:
    if cur > 0:
        busy += t - last; area += (t - last) * cur
    hist[cur] += t - last
    cur += d; last = t
print(f"span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%), sum of kernel time {area/1e6:.2f} ms, mean concurrency while busy {area/max(busy,1):.2f}")
print("time share by number of kernels in flight:", {k: f"{100*v/span:.1f}%" for k, v in sorted(hist.items())})
q = collections.defaultdict(int)
for s, e, _, qq in rows: q[qq] += e - s
print("per-queue busy share:", {k: f"{100*v/span:.1f}%" for k, v in sorted(q.items())})
k = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in rows:
    k[n][0] += e - s; k[n][1] += 1
for n, (t, c) in sorted(k.items(), key=lambda x: -x[1][0])[:16]:
    print(f"  {n:40s} calls {c:5d} mean {t/c/1e3:8.2f} us  share of span {100*t/span:5.1f}%")
