#!/usr/bin/env python3
"""Round-3 probe of the host boundary: what a pinned -> device copy costs by itself (per copy, by size, one stream / two streams),
and where the time of cox_integrate_points_async goes (time inside the calls vs the rate of the whole loop).
  python scripts/h2d_probe.py [frames] [method]          (COX_H2D=kernel|memcpy in the environment)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
method = sys.argv[2] if len(sys.argv) > 2 else "merged"
voxel = 0.05
env = {k: v for k, v in os.environ.items() if k.startswith("COX_")}
eng = coxgraph_amd.load_engine()
cfg = eng.default_config(**synth.integrator_overrides(voxel))

# ---- A: raw copies ------------------------------------------------------------------------------------------------------
if not os.environ.get("PROBE_SKIP_RAW"):
    for mb in (0.3, 1.2, 3.7, 4.9):
        nb = int(mb * 1e6) // 16 * 16
        src = [torch.empty(nb, dtype=torch.uint8).pin_memory() for _ in range(8)]
        dst = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(8)]
        for n_streams in (1, 2):
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    with torch.cuda.stream(streams[i % n_streams]):
                        dst[i % 8].copy_(src[i % 8], non_blocking=True)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            print(f"raw copy {mb} MB, {n_streams} stream(s): submit {(t1 - t0) / 200 * 1e6:.1f} us/copy, done {(t2 - t0) / 200 * 1e6:.1f} us/copy = {nb / ((t2 - t0) / 200) / 1e9:.1f} GB/s", flush=True)
        del src, dst

# ---- B: the engine ------------------------------------------------------------------------------------------------------
host = [synth.make_frame(t) for t in range(n)]
dev = [(T, torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda(), p.shape[0]) for T, p, c, _ in host]
pinned = [(T, torch.from_numpy(p).pin_memory(), torch.from_numpy(c).pin_memory()) for T, p, c, _ in host]
torch.cuda.synchronize()


def run(label, call, frames, reps=3):
    for rep in range(reps):
        integ = Integrator(eng, Layer(eng, voxel, capacity_blocks=32768), cfg, method)
        for f in frames[:10]:
            call(integ, f)
        integ.sync()
        inside = 0.0
        t0 = time.perf_counter()
        for f in frames:
            a = time.perf_counter()
            call(integ, f)
            inside += time.perf_counter() - a
        t1 = time.perf_counter()
        integ.sync()
        t2 = time.perf_counter()
        hm, hf = integ.host_time()
        print(f"{method} {label} env {env}: {len(frames) / (t2 - t0):.0f} frames/s; inside the calls {inside / len(frames) * 1e6:.0f} us/frame, loop {(t1 - t0) / len(frames) * 1e6:.0f} us/frame, "
              f"drain {(t2 - t1) * 1e6:.0f} us; engine host time {hm / max(hf, 1) * 1e3:.0f} us/frame", flush=True)
        del integ


run("resident", lambda I, f: I.integrate_points_dev(f[0], f[1].data_ptr(), f[2].data_ptr(), f[3]), dev)
run("pinned xyz+rgba", lambda I, f: I.integrate_points_async(f[0], f[1].data_ptr(), f[2].data_ptr(), f[1].shape[0]), pinned)
run("pinned xyz only", lambda I, f: I.integrate_points_async(f[0], f[1].data_ptr(), None, f[1].shape[0]), pinned)
