#!/usr/bin/env python3
"""Assemble profiles/rNN_* from one collection run on the GPU box (scripts/collect_profiles.sh; see profiles/README.md).

    python scripts/make_profiles.py gpurun_out/r2prof r02
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
HERE = os.path.dirname(os.path.abspath(__file__))
COMMON = "--cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-profile-pass --no-ramp"
FRAMES_TRACE = 320   # warmup 20 + steps 300
FRAMES_PMC = 40      # the last 40 of the 80 frames (--warmup 20 --steps 60 --no-ramp): steady state


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", f"*{suffix}"), recursive=True)
    if not hits:
        raise SystemExit(f"missing {suffix} under {src}/{sub}")
    return hits[0]


def run(*a):
    return subprocess.run([sys.executable, *a], capture_output=True, text=True, check=True).stdout


def copy(a, b):
    open(b, "w").write(open(a).read())


commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
copy(os.path.join(src, "bench_line.json"), os.path.join(P, f"{tag}_bench_line.json"))
if os.path.exists(os.path.join(src, "bench_line_driver.json")):
    copy(os.path.join(src, "bench_line_driver.json"), os.path.join(P, f"{tag}_bench_line_driver.json"))
for m in ("merged", "fast"):
    for mode, flag, what in (("async", "", "frames overlapped as in the timed run"), ("serial", " --serial", "one frame in flight")):
        stats = find(f"{m}_{mode}", "kernel_stats.csv")
        name = f"{tag}_bench_{m}{'_serial' if mode == 'serial' else ''}_kernel_stats"
        copy(stats, os.path.join(P, name + ".csv"))
        head = f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --method {m} {COMMON}{flag}   ({what}; us/frame = total / {FRAMES_TRACE} frames: 20 warm-up + 300 timed)"
        open(os.path.join(P, name + ".txt"), "w").write(head + "\n" + run(os.path.join(HERE, "kstats.py"), stats, str(FRAMES_TRACE)))
# fine voxels: one-frame-in-flight kernel statistics + the bench line of the same configuration
for vox, frames in (("2cm", 50), ("1cm", 40)):
    try:
        stats = find(f"merged_serial_{vox}", "kernel_stats.csv")
        name = f"{tag}_bench_merged_serial_{vox}_kernel_stats"
        copy(stats, os.path.join(P, name + ".csv"))
        head = (f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --method merged --voxel 0.0{vox[0]} --steps {frames - 10} --warmup 10 --serial --no-events "
                f"--cpu-frames 0 --reg-iters 0 --other-frames 0 --pcie-frames 0 --no-profile-pass --no-ramp   (us/frame = total / {frames} frames)")
        open(os.path.join(P, name + ".txt"), "w").write(head + "\n" + run(os.path.join(HERE, "kstats.py"), stats, str(frames)))
        copy(os.path.join(src, f"bench_line_{vox}.json"), os.path.join(P, f"{tag}_bench_line_{vox}.json"))
    except (SystemExit, FileNotFoundError) as e:
        print("fine-voxel profile skipped:", vox, e)
if os.path.exists(os.path.join(src, "bench_line_10cm.json")):
    copy(os.path.join(src, "bench_line_10cm.json"), os.path.join(P, f"{tag}_bench_line_10cm.json"))
for extra in ("stage_timeline_5cm.txt",):
    if os.path.exists(os.path.join(src, extra)):
        copy(os.path.join(src, extra), os.path.join(P, f"{tag}_{extra}"))
try:
    open(os.path.join(P, f"{tag}_timeline_concurrency.txt"), "w").write(
        f"# kernels in flight during the middle half of the fusion frames of: rocprofv3 --kernel-trace -- python3 bench.py --method merged {COMMON}\n" +
        run(os.path.join(HERE, "timeline.py"), find("merged_async", "kernel_trace.csv")))
except Exception as e:  # the timeline is a nice-to-have
    print("timeline skipped:", e)


STEADY = 40  # frames of the PMC run that are counted: the last 40 (set-up -- a 1.6 GB pool memset, the first frames' block allocations -- stays out)


def pmc(path, name):
    """Per kernel: summed counter and dispatch count over the LAST `STEADY` frames of the run.  A frame ends with its
    apply dispatch (k_apply_block by default: exactly one per frame, the last kernel of a frame in --serial mode)."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"].split("(")[0] for k in ("k_apply_block", "k_apply_pieces", "k_apply_long"))]
    assert len(ends) >= STEADY + 1, len(ends)
    lo, hi = ends[-STEADY - 1] + 1, ends[-1] + 1
    acc, cnt, order = collections.defaultdict(float), collections.defaultdict(int), []
    for r in rows[lo:hi]:
        k = r["Kernel_Name"].split("(")[0]
        if k not in acc:
            order.append(k)
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return acc, cnt, order


for m in ("merged", "fast"):
    cmd = f"python3 bench.py --method {m} --serial --steps 60 --warmup 20 --cpu-frames 0 --reg-iters 0 --other-frames 0 --other-config-frames 0 --pcie-frames 0 --no-profile-pass --no-events --no-ramp"
    f, fc, order = pmc(find(f"{m}_pmc_fetch", "counter_collection.csv"), "FETCH_SIZE")
    w, wc, _ = pmc(find(f"{m}_pmc_write", "counter_collection.csv"), "WRITE_SIZE")
    lines = [f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE (another pass) -- {cmd}",
             f"# per kernel: launches per frame, KB per launch (raw counter), MB per frame; {FRAMES_PMC} frames",
             f"{'kernel':46s} {'launch/frame':>12s} {'FETCH KB':>10s} {'WRITE KB':>10s} {'MB/frame':>9s}"]
    total, launches = 0.0, 0.0
    per_kernel = {}
    for k in order:
        lf = fc[k] / FRAMES_PMC
        fk, wk = f[k] / fc[k], (w.get(k, 0.0) / wc[k] if wc.get(k) else 0.0)
        mb = (f[k] + w.get(k, 0.0)) * 1024 / FRAMES_PMC / 1e6
        total += mb
        launches += lf
        per_kernel[k] = {"launches_per_frame": lf, "fetch_bytes_per_launch": fk * 1024, "write_bytes_per_launch": wk * 1024, "bytes_per_frame": mb * 1e6}
        lines.append(f"{k[:46]:46s} {lf:12.2f} {fk:10.0f} {wk:10.0f} {mb:9.2f}")
    lines.append(f"{'TOTAL':46s} {launches:12.2f} {'':>10s} {'':>10s} {total:9.2f}")
    open(os.path.join(P, f"{tag}_pmc_traffic_{m}.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- {cmd}",
               "commit": commit, "frames": FRAMES_PMC, "unit": "bytes", "bytes_per_frame": total * 1e6, "launches_per_frame": launches,
               "note": "sum over ALL kernels dispatched during the last 40 frames of the run of (FETCH_SIZE + WRITE_SIZE) x 1024 / 40 (raw counters in KB; set-up such as the 1.6 GB pool memset stays out); FETCH_SIZE is NOT doubled: the gfx950 "
                       "half-count applies to 16-B/lane streaming reads, these kernels gather 4-12 B/lane (uncalibrated, MI355X_MICROARCH.md HBM section)",
               "kernels": per_kernel}, open(os.path.join(P, f"{tag}_pmc_traffic_{m}.json"), "w"), indent=1)
print("profiles written for", tag, "commit", commit)
