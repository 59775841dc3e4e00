import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import coxgraph_amd
from coxgraph_amd import synth
from coxgraph_amd.capi import Layer, Integrator
voxel = float(sys.argv[1]) if len(sys.argv) > 1 else 0.10
ramp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = coxgraph_amd.load_engine()
cfg = eng.default_config(**synth.integrator_overrides(voxel))
frames = []
for t in range(110):
    T, p, c, _ = synth.make_frame(t)
    frames.append((T, torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda(), p.shape[0]))
torch.cuda.synchronize()
def run(I, fr):
    for T, x, c, k in fr:
        I.integrate_points_dev(T, x.data_ptr(), c.data_ptr(), k)
A = Integrator(eng, Layer(eng, voxel, capacity_blocks=32768), cfg, "merged")
if ramp == 1:      # bench.py's clock ramp: a second integrator, created after A, used for 0.3 s, destroyed
    B = Integrator(eng, Layer(eng, voxel, capacity_blocks=32768), cfg, "merged")
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.3:
        run(B, frames[i % 100:i % 100 + 1]); i += 1
        if i % 16 == 0: B.sync()
    B.sync(); del B
elif ramp == 2:    # the same ramp on A itself
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.3:
        run(A, frames[i % 100:i % 100 + 1]); i += 1
        if i % 16 == 0: A.sync()
    A.sync()
run(A, frames[:10]); A.sync()
for rep in range(4):
    t0 = time.perf_counter(); run(A, frames[10:110]); A.sync(); dt = time.perf_counter() - t0
    print(f"voxel {voxel} ramp {ramp} rep {rep}: {100 / dt:.0f} frames/s", flush=True)
