#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel_stats.csv as a table (per-frame figures if --frames given)."""
import csv, sys, glob
path = sys.argv[1]
frames = float(sys.argv[2]) if len(sys.argv) > 2 else None
f = glob.glob(path + "/*/*kernel_stats.csv")[0] if not path.endswith(".csv") else path
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}" + ("  us/frame" if frames else ""))
for r in rows[:40]:
    t = float(r["TotalDurationNs"])
    line = f"{r['Name'].split('(')[0][:60]:60s} {r['Calls']:>7s} {t/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f} {100*t/tot:6.2f}"
    if frames:
        line += f"  {t/1e3/frames:8.1f}"
    print(line)
print(f"total kernel time {tot/1e6:.3f} ms" + (f" = {tot/1e3/frames:.1f} us/frame" if frames else ""))
