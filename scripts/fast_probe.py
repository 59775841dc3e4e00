#!/usr/bin/env python3
"""Round-3 probe: the benchmark stream through `fast` (or another method), frames resident in HBM, enqueued back to back;
prints frames/s and the relaxation's run totals.   python scripts/fast_probe.py [frames] [voxel] [method]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
method = sys.argv[3] if len(sys.argv) > 3 else "fast"
eng = coxgraph_amd.load_engine()
cfg = eng.default_config(integrator_threads=1, **synth.integrator_overrides(voxel))
frames = []
for t in range(n):
    T, p, c, _ = synth.make_frame(t)
    frames.append((T, torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda(), p.shape[0]))
torch.cuda.synchronize()
for rep in range(2):
    layer = Layer(eng, voxel, capacity_blocks=32768)
    integ = Integrator(eng, layer, cfg, method)
    for T, x, c, k in frames[:10]:
        integ.integrate_points_dev(T, x.data_ptr(), c.data_ptr(), k)
    integ.sync()
    t0 = time.perf_counter()
    for T, x, c, k in frames[10:]:
        integ.integrate_points_dev(T, x.data_ptr(), c.data_ptr(), k)
    integ.sync()
    dt = time.perf_counter() - t0
    st = integ.fast_stats() if method == "fast" else {}
    hm, hf = integ.host_time()
    if rep == 1 and method == "fast":  # class times, one frame in flight
        integ.set_profiling(True)
        integ.class_times(reset=True)
        for T, x, c, k in frames[10:50]:
            integ.integrate_points_dev(T, x.data_ptr(), c.data_ptr(), k)
            integ.sync()
        ct = integ.class_times()
        print("  class ms/frame (one frame in flight):", {k: round(v[0] / 40, 4) for k, v in ct.items() if v[1]}, flush=True)
    print(f"{method} voxel {voxel} env {({k: v for k, v in os.environ.items() if k.startswith('COX_')})}: {(n - 10) / dt:.0f} frames/s, host {hm / max(hf, 1):.3f} ms/frame, {st}", flush=True)
    del integ, layer
