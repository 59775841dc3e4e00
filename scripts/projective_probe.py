#!/usr/bin/env python3
"""Frames/s and HBM bytes of the projective integrator on the benchmark's stream (GPU box).  python scripts/projective_probe.py [voxel]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator

voxel = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
eng = coxgraph_amd.load_engine()
ov = synth.integrator_overrides(voxel)
cfg = eng.default_config(sensor_horizontal_resolution=1280, sensor_vertical_resolution=960, sensor_vertical_field_of_view_degrees=360.0,
                         default_truncation_distance=ov["default_truncation_distance"], min_ray_length_m=ov["min_ray_length_m"], max_ray_length_m=ov["max_ray_length_m"],
                         use_const_weight=1)
frames = [synth.make_frame(t) for t in range(n)]
dev = [(T, torch.from_numpy(p).cuda()) for T, p, _, _ in frames]
layer = Layer(eng, voxel, capacity_blocks=32768)
integ = Integrator(eng, layer, cfg, "projective")
for T, xyz in dev[:20]:
    integ.integrate_points_dev(T, xyz.data_ptr(), 0, xyz.shape[0])
integ.sync()
torch.cuda.synchronize()
t0 = time.perf_counter()
upd = 0
for T, xyz in dev[20:]:
    integ.integrate_points_dev(T, xyz.data_ptr(), 0, xyz.shape[0])
integ.sync()
dt = time.perf_counter() - t0
st = integ.last_stats()
if len(sys.argv) > 3:  # a second pass, one frame at a time: mean marked blocks per frame (for the update kernel's bytes per launch)
    layer2 = Layer(eng, voxel, capacity_blocks=32768)
    integ2 = Integrator(eng, layer2, cfg, "projective")
    tb = []
    for T, xyz in dev:
        integ2.integrate_points_dev(T, xyz.data_ptr(), 0, xyz.shape[0])
        tb.append(integ2.last_stats()["n_touched_blocks"])
    print(f"marked blocks per frame: mean {np.mean(tb):.1f} (min {min(tb)}, max {max(tb)}) -> k_proj_update streams {np.mean(tb) * 2 * 49152 / 1e6:.2f} MB per launch on average")
print(f"voxel {voxel}: {(n - 20) / dt:.1f} frames/s, {dt / (n - 20) * 1e3:.3f} ms/frame; last frame: {st}; blocks {layer.stats()[0]}")
