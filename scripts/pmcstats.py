#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (counter_collection.csv)."""
import csv, sys, glob, collections
path = sys.argv[1]
f = glob.glob(path + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
print(f"{'kernel':40s} " + " ".join(f"{n[-16:]:>16s}" for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print(f"{k:40s} " + " ".join(f"{acc[k][n]/max(cnt[k][n],1):16.0f}" for n in names))
