#!/usr/bin/env python3
"""Assemble profiles/rNN_* from one collection run on the GPU box (see profiles/README.md for the commands).

    python scripts/make_profiles.py gpurun_out/r r01
"""
import collections
import csv
import json
import os
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
HERE = os.path.dirname(os.path.abspath(__file__))


def run(*a):
    return subprocess.run([sys.executable, *a], capture_output=True, text=True, check=True).stdout


def copy(a, b):
    open(b, "w").write(open(a).read())


CMD = "python bench.py --cpu-frames 0 --reg-iters 8 --fast-frames 0 --no-profile-pass"
for sub, pre, name, head in (
        ("async", "a", "bench", f"# rocprofv3 --kernel-trace --stats -- {CMD}   (merged, 320 frames, frames overlapped on two streams)"),
        ("serial", "s", "bench_serial", f"# rocprofv3 --kernel-trace --stats -- {CMD} --serial   (one frame in flight)"),
        ("fast", "f", "bench_fast", "# rocprofv3 --kernel-trace --stats -- python bench.py --method fast --cpu-frames 0 --reg-iters 8 --no-profile-pass   (320 frames)")):
    stats = os.path.join(src, sub, f"{pre}_kernel_stats.csv")
    copy(stats, os.path.join(P, f"{tag}_{name}_kernel_stats.csv"))
    open(os.path.join(P, f"{tag}_{name}_kernel_stats.txt"), "w").write(head + "\n" + run(os.path.join(HERE, "kstats.py"), stats, "320"))
open(os.path.join(P, f"{tag}_timeline_concurrency.txt"), "w").write(
    f"# kernels in flight during the middle half of the fusion frames of: rocprofv3 --kernel-trace -- {CMD}\n" +
    run(os.path.join(HERE, "timeline.py"), os.path.join(src, "async", "a_kernel_trace.csv")))
copy(os.path.join(src, "bench_line.json"), os.path.join(P, f"{tag}_bench_line.json"))


def pmc(path, name):
    acc, cnt, order = collections.defaultdict(float), collections.defaultdict(int), []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        if k not in acc:
            order.append(k)
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in order}, order


PMC_CMD = "python bench.py --serial --steps 60 --warmup 20 --cpu-frames 0 --reg-iters 0 --fast-frames 0 --no-profile-pass --no-events"
f, order = pmc(os.path.join(src, "pmc_fetch", "p_counter_collection.csv"), "FETCH_SIZE")
w, _ = pmc(os.path.join(src, "pmc_write", "p_counter_collection.csv"), "WRITE_SIZE")
lines = [f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE (another pass) -- {PMC_CMD}; mean per launch, KB",
         f"{'kernel':46s} {'FETCH_SIZE':>12s} {'WRITE_SIZE':>12s}"]
lines += [f"{k[:46]:46s} {f[k]:12.0f} {w.get(k, float('nan')):12.0f}" for k in order]
open(os.path.join(P, f"{tag}_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- {PMC_CMD}; mean per launch",
           "unit": "bytes",
           "note": "raw counter x 1024 (KB); FETCH_SIZE is NOT doubled: the gfx950 half-count applies to 16-B/lane streaming reads, these "
                   "kernels gather 4-12 B/lane (uncalibrated, MI355X_MICROARCH.md HBM section)",
           "kernels": {k: {"fetch_bytes": f[k] * 1024, "write_bytes": w.get(k, 0.0) * 1024} for k in order}},
          open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("profiles written for", tag)
